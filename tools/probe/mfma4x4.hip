// Operand layout probe of v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4x1) on gfx950: which (A lane, B lane) product lands in which
// (lane, register) of the result, plain and with the A-matrix broadcast (cbsz = 4: all 16 blocks take block `abid`'s A lanes).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma4x4.hip -o tools/probe/mfma4x4 && ./tools/probe/mfma4x4
// Result (MI355X): D[lane 4b+j][reg i] = A[lane 4b+i] * B[lane 4b+j];  cbsz=4, abid=a: D[lane 4b+j][reg i] = A[lane 4a+i] * B[lane 4b+j].
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, CBSZ, ABID, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
static void report(const char* what, const float* ha, const float* hb, const float* hd) {
    printf("%s\n", what);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int fa = -1, fb = -1, n = 0;
        for (int x = 0; x < 64; ++x) for (int y = 0; y < 64; ++y)
            if (fabsf(ha[x] * hb[y] - hd[l * 4 + r]) < 2e-7f * fabsf(hd[l * 4 + r])) { fa = x; fb = y; ++n; }
        if (l < 6 || l == 21 || l == 63) printf("  D[lane %2d][reg %d] = A[lane %2d] * B[lane %2d]   (%d match)\n", l, r, fa, fb, n);
    }
}
int main() {
    float ha[64], hb[64], hd[256];
    for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i * 0.3712f + (i * i % 7) * 0.0113f; hb[i] = 2.0f + i * 0.5371f + (i * i % 5) * 0.0131f; }
    float *a, *b, *d;
    (void)hipMalloc(&a, 256); (void)hipMalloc(&b, 256); (void)hipMalloc(&d, 1024);
    (void)hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    report("cbsz=0", ha, hb, hd);
    hipLaunchKernelGGL((k<4, 0>), dim3(1), dim3(64), 0, 0, a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    report("cbsz=4 abid=0", ha, hb, hd);
    hipLaunchKernelGGL((k<4, 5>), dim3(1), dim3(64), 0, 0, a, b, d);
    (void)hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
    report("cbsz=4 abid=5", ha, hb, hd);
    return 0;
}
