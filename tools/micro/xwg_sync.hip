// How long does one all-to-all exchange between G workgroups of ONE kernel take through global memory (agent-scope
// release / acquire)?  Each round every workgroup publishes a 64-bit value tagged with the round and waits until it has
// seen all G values of that round - the per-round arg-max exchange a multi-workgroup FPS would need.  Spins are BOUNDED.
// Build: hipcc --offload-arch=gfx950 -O3 xwg_sync.hip -o xwg_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// stride 1: workgroups 0..G-1 (dealt round-robin over the 8 XCDs); stride 8: workgroups 0, 8, 16, ... (one XCD, if the
// dispatcher deals in order) - the others exit at once
__global__ __launch_bounds__(64) void exchange(unsigned long long* slots, int G, int rounds, unsigned long long* out, int* fail, int stride) {
    if (blockIdx.x % stride) return;
    const int g = blockIdx.x / stride, lane = threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        unsigned long long* row = slots + (size_t)(r % 3) * G;
        if (lane == 0) __hip_atomic_store(row + g, ((unsigned long long)r << 32) | (unsigned)(g * 7 + r), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v = 0;
        int spins = 0;
        bool ok;
        do {
            v = (lane < G) ? __hip_atomic_load(row + lane, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)r << 32);
            ok = __all((int)(v >> 32) == r);
        } while (!ok && ++spins < 2000000);
        if (!ok) { if (lane == 0) *fail = 1; return; }
        unsigned long long m = v;
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
        acc += m;
    }
    if (lane == 0) out[g] = acc;
}

int main() {
    for (int stride : {1, 8})
    for (int G : {2, 4, 8, 16, 32}) {
        unsigned long long *slots, *out; int* fail;
        hipMalloc(&slots, 3 * 64 * 8); hipMemset(slots, 0, 3 * 64 * 8);
        hipMalloc(&out, 64 * 8); hipMalloc(&fail, 4); hipMemset(fail, 0, 4);
        const int rounds = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        exchange<<<G * stride, 64>>>(slots, G, 100, out, fail, stride);
        hipDeviceSynchronize(); hipMemset(slots, 0, 3 * 64 * 8);
        hipEventRecord(e0);
        exchange<<<G * stride, 64>>>(slots, G, rounds, out, fail, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("stride %d  G = %2d workgroups: %.0f ns per exchange round%s\n", stride, G, ms * 1e6 / rounds, f ? "  (TIMEOUT)" : "");
        hipFree(slots); hipFree(out); hipFree(fail);
    }
    return 0;
}
