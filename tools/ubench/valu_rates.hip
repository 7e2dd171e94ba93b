// valu_rates.hip -- issue rates of the instructions the posting-scan kernels are made of, on gfx950 (MI355X).
// Each kernel runs ITER x 64 instructions of ONE kind per wave (8 independent dependency chains), on 256 CUs x 4 SIMDs x
// W waves per SIMD; prints cycles per wave-instruction per SIMD (kernel cycles x SIMDs x ... / instructions).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip      Run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ITER = 2000;

#define OP8(fmt) \
    asm volatile(fmt : "+v"(a0) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a2) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a4) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(fmt : "+v"(a6) : "v"(b), "v"(c)); asm volatile(fmt : "+v"(a7) : "v"(b), "v"(c));
#define OP64(fmt) OP8(fmt) OP8(fmt) OP8(fmt) OP8(fmt) OP8(fmt) OP8(fmt) OP8(fmt) OP8(fmt)

#define KERNEL32(name, fmt)                                                                 \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) {              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = seed | 1u, c = seed ^ 0x55u;                                            \
        for (int i = 0; i < ITER; ++i) { OP64(fmt) }                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
    }
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_lshl, "v_lshlrev_b32 %0, %1, %0")
KERNEL32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL32(k_bfe, "v_bfe_u32 %0, %0, 5, 12")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL32(k_mullo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cmp, "v_cmp_lt_u32 vcc, %0, %1")
KERNEL32(k_dpp, "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL32(k_fma32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_readlane_like, "v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_and, "v_and_b32 %0, %0, %1")
KERNEL32(k_or, "v_or_b32 %0, %0, %1")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_bfi, "v_bfi_b32 %0, %0, %1, %2")
KERNEL32(k_min3, "v_min3_u32 %0, %0, %1, %2")
KERNEL32(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL32(k_cmp_cnd, "v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_cmp64_cnd, "v_cmp_lt_u32 s[20:21], %0, %1\n\tv_cndmask_b32 %0, %0, %2, s[20:21]")
KERNEL32(k_cnd_s, "v_cndmask_b32 %0, %0, %1, s[22:23]")
KERNEL32(k_readlane, "v_readlane_b32 s24, %0, 3")
KERNEL32(k_readfirst, "v_readfirstlane_b32 s24, %0")
KERNEL32(k_writelane, "v_writelane_b32 %0, s4, 5")
KERNEL32(k_bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL32(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL32(k_cvt_f64_u32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_sub, "v_sub_u32 %0, %0, %1")
KERNEL32(k_lshr, "v_lshrrev_b32 %0, %1, %0")
KERNEL32(k_min, "v_min_u32 %0, %0, %1")

#define OP8D(fmt) \
    asm volatile(fmt : "+v"(d0) : "v"(e), "v"(f)); asm volatile(fmt : "+v"(d1) : "v"(e), "v"(f)); \
    asm volatile(fmt : "+v"(d2) : "v"(e), "v"(f)); asm volatile(fmt : "+v"(d3) : "v"(e), "v"(f)); \
    asm volatile(fmt : "+v"(d4) : "v"(e), "v"(f)); asm volatile(fmt : "+v"(d5) : "v"(e), "v"(f)); \
    asm volatile(fmt : "+v"(d6) : "v"(e), "v"(f)); asm volatile(fmt : "+v"(d7) : "v"(e), "v"(f));
#define OP64D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt) OP8D(fmt)
#define KERNEL64(name, fmt)                                                                  \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t seed) {               \
        double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7; \
        double e = 1.0 + seed * 1e-9, f = 1e-9;                                               \
        for (int i = 0; i < ITER; ++i) { OP64D(fmt) }                                         \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7); \
    }
KERNEL64(k_fma64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_add64, "v_add_f64 %0, %0, %1")
KERNEL64(k_mul64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_rcp64, "v_rcp_f64 %0, %0")

// the IEEE f64 divide the compiler emits for a / b (what Cache::evaluate costs)
__global__ void __launch_bounds__(256) k_div64(uint32_t *out, uint32_t seed) {
    double d0 = threadIdx.x + 1.0, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;
    const double e = 1.0 + seed * 1e-9;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            d0 = d0 / e + 1.0; d1 = d1 / e + 1.0; d2 = d2 / e + 1.0; d3 = d3 / e + 1.0;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(d0 + d1 + d2 + d3);
}

// LDS: mode 0 ds_or_rtn_b32 at random words, 1 ds_read_b32 random, 2 ds_read2_b64 (two 8-byte reads 16 bytes apart) linear,
// 3 ds_write_b64 linear, 4 ds_or (no return) random
template <int MODE>
__global__ void __launch_bounds__(256) k_lds(uint32_t *out, uint32_t seed) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = i * seed;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + seed, acc = 0;
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            x = x * 1664525u + 1013904223u;
            const uint32_t w = (x >> 12) & 4095u;
            if (MODE == 0) acc += atomicOr(&s[w], 1u << (x & 31));
            if (MODE == 1) acc += s[w];
            if (MODE == 2) {
                const uint2 *p = reinterpret_cast<const uint2 *>(&s[((threadIdx.x & 63) * 4 + (j & 7) * 256) & 8188]);
                const uint2 a = p[0], b2 = p[2];
                acc += a.x + a.y + b2.x + b2.y;
            }
            if (MODE == 3) *reinterpret_cast<uint2 *>(&s[((threadIdx.x * 2) + (j & 7) * 512) & 8190]) = make_uint2(x, acc);
            if (MODE == 4) atomicOr(&s[w], 1u << (x & 31));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + s[threadIdx.x];
}

template <class K>
static int run(const char *name, K kern, int insts_per_iter, uint32_t *out) {
    for (int wps : {1, 2, 4}) {  // waves per SIMD: 256-thread blocks = 1 wave per SIMD each
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        kern<<<blocks, 256>>>(out, 12345u);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        kern<<<blocks, 256>>>(out, 12345u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double per_simd = (double)ITER * insts_per_iter * wps;   // wave-instructions issued on one SIMD
        const double ns_per = ms * 1e6 / per_simd;
        printf("%-16s waves/SIMD %d: %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", name, wps, ms, ns_per, ns_per * 2.4);
    }
    return 0;
}

int main() {
    uint32_t *out;
    CHK(hipMalloc(&out, 4 * 256 * 4 * 256));
#define R(k, n) if (run(#k, k, n, out)) return 1;
    R(k_add, 64) R(k_sub, 64) R(k_xor, 64) R(k_lshl, 64) R(k_lshr, 64) R(k_and_or, 64) R(k_alignbit, 64) R(k_add3, 64) R(k_lshl_add, 64)
    R(k_bfe, 64) R(k_min, 64) R(k_mul24, 64) R(k_mad24, 64) R(k_mullo, 64) R(k_cndmask, 64) R(k_cmp, 64) R(k_dpp, 64) R(k_readlane_like, 64)
    R(k_and, 64) R(k_or, 64) R(k_mov, 64) R(k_bfi, 64) R(k_min3, 64) R(k_lshl_or, 64) R(k_cmp_cnd, 128) R(k_cmp64_cnd, 128) R(k_cnd_s, 64)
    R(k_readlane, 64) R(k_readfirst, 64) R(k_writelane, 64) R(k_bcnt, 64) R(k_perm, 64) R(k_cvt_f64_u32, 64)
    R(k_fma32, 64) R(k_fma64, 64) R(k_add64, 64) R(k_mul64, 64) R(k_rcp64, 64)
    if (run("k_div64(a/b+1)", k_div64, 64, out)) return 1;
    if (run("lds or_rtn rand", k_lds<0>, 16, out)) return 1;
    if (run("lds read_b32 rand", k_lds<1>, 16, out)) return 1;
    if (run("lds read2_b64", k_lds<2>, 16, out)) return 1;
    if (run("lds write_b64", k_lds<3>, 16, out)) return 1;
    if (run("lds or rand", k_lds<4>, 16, out)) return 1;
    return 0;
}
