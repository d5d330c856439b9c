// Training-batch augmentation on the device (DisPU/dataset.py:118-143 applies, on the host in numpy, in this order:
// jitter -> rotate -> scale; Common/point_operation.py:32-71,73-85,87-104,107-123).  One pass over the batch:
//   out[b,i,:] = ((in[b,i,:] + noise[b,i,:]) . R_b) * scale_b + shift_b
// The random draws (clipped Gaussian jitter, angles, scales, shifts) are made by the host with the reference's own
// numpy call sequence and handed in, so a seeded run reproduces the reference's batches; the HBM-bound transform of
// the [B, N, 3] coordinate blocks happens here.  noise / shift may be NULL.
#include "common.h"

namespace dispu {

__global__ void augment_kernel(int b, int n, const float* __restrict__ in, const float* __restrict__ noise,
                               const float* __restrict__ rot, const float* __restrict__ scale, const float* __restrict__ shift,
                               float* __restrict__ out) {
    const size_t total = (size_t)b * n;
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t c = e / n;
        float x = in[e * 3 + 0], y = in[e * 3 + 1], z = in[e * 3 + 2];
        if (noise) { x += noise[e * 3 + 0]; y += noise[e * 3 + 1]; z += noise[e * 3 + 2]; }
        const float* R = rot + c * 9;                       // row vector times matrix: p' = p . R (np.dot(points, R))
        float ox = (x * R[0] + y * R[3]) + z * R[6];
        float oy = (x * R[1] + y * R[4]) + z * R[7];
        float oz = (x * R[2] + y * R[5]) + z * R[8];
        const float s = scale[c];
        ox *= s; oy *= s; oz *= s;
        if (shift) { ox += shift[c * 3 + 0]; oy += shift[c * 3 + 1]; oz += shift[c * 3 + 2]; }
        out[e * 3 + 0] = ox; out[e * 3 + 1] = oy; out[e * 3 + 2] = oz;
    }
}

}  // namespace dispu

using namespace dispu;

DISPU_EXPORT int dispu_augment(int b, int n, const float* in, const float* noise, const float* rot, const float* scale,
                               const float* shift, float* out, void* stream) {
    if (b < 0 || n < 0 || !rot || !scale) return (int)hipErrorInvalidValue;
    if (b == 0 || n == 0) return 0;
    const size_t total = (size_t)b * n;
    const size_t g = (total + 255) / 256;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)(g > 8192 ? 8192 : g)), dim3(256), 0, (hipStream_t)stream, b, n, in, noise, rot,
                       scale, shift, out);
    return (int)hipGetLastError();
}
