// Which CUs does a CU-masked stream (hipExtStreamCreateWithCUMask) use on MI355X, per XCD?  Prints, for masks of the low / high
// half of the bits and for even / odd bits, the number of distinct (XCC, SE, CU) slots a 2048-workgroup launch touched per XCC.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <set>
#include <vector>
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void who(unsigned* out) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xF; out[2 * blockIdx.x + 1] = hwid; }
    // stay a little so that the launch spreads over every admitted CU
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    HIPC(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int nb = 4096;
    unsigned* d; HIPC(hipMalloc(&d, nb * 8));
    who<<<nb, 256, 65536, s>>>(d);     // 64 KB of LDS: at most two workgroups per CU
    HIPC(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * nb);
    HIPC(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
    std::set<unsigned> per[16];
    for (int i = 0; i < nb; ++i) {
        unsigned xcc = h[2 * i], hw = h[2 * i + 1];
        unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        per[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%-28s:", name);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" xcc%d %2zu", x, per[x].size()); tot += (int)per[x].size(); }
    printf("  total %d CUs\n", tot);
    HIPC(hipFree(d)); HIPC(hipStreamDestroy(s));
}

int main() {
    hipDeviceProp_t pr; HIPC(hipGetDeviceProperties(&pr, 0));
    printf("%s: %d CUs\n", pr.gcnArchName, pr.multiProcessorCount);
    std::vector<uint32_t> all(8, 0xFFFFFFFFu), lo(8, 0), hi(8, 0), even(8, 0x55555555u), odd(8, 0xAAAAAAAAu), first64(8, 0), x8(8, 0);
    for (int i = 0; i < 4; ++i) { lo[i] = 0xFFFFFFFFu; hi[4 + i] = 0xFFFFFFFFu; }
    first64[0] = first64[1] = 0xFFFFFFFFu;
    for (int i = 0; i < 256; ++i) if ((i / 8) % 2 == 0) x8[i / 32] |= 1u << (i % 32);   // alternating groups of 8 bits
    run("all 256 bits", all);
    run("bits 0..127", lo);
    run("bits 128..255", hi);
    run("even bits", even);
    run("odd bits", odd);
    run("bits 0..63", first64);
    run("alternating groups of 8", x8);
    // Round 6: can a mask select WHOLE XCDs?  bit i -> XCD i mod 8 by the rows above, so "i mod 8 < 4" asks for
    // XCDs 0-3 with all their CUs and none on XCDs 4-7, the complement for the other four; the third mask keeps one CU-pair row alive
    // on the excluded XCDs (bits 4..7 of the first group of eight).
    std::vector<uint32_t> x03(8, 0), x47(8, 0), x03p(8, 0);
    for (int i = 0; i < 256; ++i) {
        if (i % 8 < 4) { x03[i / 32] |= 1u << (i % 32); x03p[i / 32] |= 1u << (i % 32); }
        else x47[i / 32] |= 1u << (i % 32);
        if (i < 8 && i % 8 >= 4) x03p[i / 32] |= 1u << (i % 32);
    }
    run("bits i mod 8 < 4 (XCD 0-3)", x03);
    run("bits i mod 8 >= 4 (XCD 4-7)", x47);
    run("XCD 0-3 + bit 4..7", x03p);
    return 0;
}
