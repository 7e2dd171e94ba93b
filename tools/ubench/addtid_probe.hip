// addtid_probe: where does ds_write_addtid_b32 write?  (address = M0[15:0] + 16-bit offset + 4 * lane, MI355X_MICROARCH.md, LDS.)
// A workgroup of 4 waves with 160 KB of LDS filled with a marker; wave `wv` issues one ds_write_addtid_b32 with a given M0 and offset;
// the whole LDS is then scanned for the words that changed.  Prints, per case, the byte address of the first changed word, the
// number of changed words, and the value found there (the lane that wrote it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int WORDS = 40960 - 256;  // 159 KB (a little room for the compiler)
template <int OFF>
__global__ void __launch_bounds__(256) probe(uint32_t m0v, int wv, uint32_t *out) {
    __shared__ uint32_t L[WORDS];
    for (int i = threadIdx.x; i < WORDS; i += 256) L[i] = 0xdeadbeefu;
    __syncthreads();
    if ((int)(threadIdx.x >> 6) == wv) {
        const uint32_t val = threadIdx.x;
        const uint32_t sm0 = __builtin_amdgcn_readfirstlane(m0v);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tds_write_addtid_b32 %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : : "s"(sm0), "v"(val), "n"(OFF) : "m0", "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t first = 0xffffffffu, n = 0, v = 0, last = 0;
        for (int i = 0; i < WORDS; ++i)
            if (L[i] != 0xdeadbeefu) {
                if (first == 0xffffffffu) { first = 4u * i; v = L[i]; }
                last = 4u * i;
                ++n;
            }
        out[0] = first; out[1] = n; out[2] = v; out[3] = last;
    }
}
template <int OFF>
void run(uint32_t m0v, int wv, uint32_t *d) {
    uint32_t h[4];
    probe<OFF><<<1, 256>>>(m0v, wv, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("M0 = 0x%08x offset = 0x%04x wave %d: expected 0x%05x; first changed 0x%05x, last 0x%05x, %u words, first value %u\n", m0v, OFF, wv,
           (m0v & 0xffffu) + OFF, h[0], h[3], h[1], h[2]);
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 16);
    run<0>(0x0000, 0, d);
    run<0>(0x2000, 1, d);
    run<256>(0x2000, 2, d);
    run<0>(0xe000, 3, d);
    run<0x2000>(0xe000, 0, d);   // sum 0x10000: beyond 64 KB
    run<0xe000>(0x2000, 1, d);   // same sum, the other way round
    run<0xe000>(0x6000, 2, d);   // 0x14000
    run<0xff00>(0xc000, 3, d);   // 0x1bf00
    run<0xe000>(0x1a000 - 0xe000, 0, d);  // 0x1a000
    run<0>(0x12000, 1, d);       // M0 above 16 bits: truncated to 0x2000?
    return 0;
}
