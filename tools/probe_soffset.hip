// Hardware probe (gfx950): is the SGPR offset of a raw buffer load part of the descriptor's range check?
//   gemm_nt_c2.hip / the weight-gradient fast path put a tile's ROW offset into soffset and rely on "rows past M read as zeros".
//   The ISA documents describe raw-buffer range checking on (inst_offset + voffset) only.  This program settles it on the machine:
//   a 4 KiB descriptor inside a 64 KiB allocation filled with 0xA5A5A5A5; loads whose voffset is in range but whose voffset + soffset is not.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_soffset.hip -o build/probe_soffset
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t* base, uint32_t records, uint32_t voff, uint32_t soff, uint32_t* out) {
    const uint64_t b = (uint64_t)(uintptr_t)base;
    i32x4 srd; srd[0] = (int)(uint32_t)b; srd[1] = (int)(uint32_t)((b >> 32) & 0xffffu); srd[2] = (int)records; srd[3] = 0x00020000;
    u32x4 v;
    const uint32_t vo = voff + threadIdx.x * 16u;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(vo), "s"(srd), "s"(soff) : "memory");
    // the same through the LDS-DMA form the GEMMs use
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xDEADBEEFu;
    __syncthreads();
    const uint32_t ldsa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_waitcnt vmcnt(0)" :: "v"(vo), "s"(srd), "s"(soff), "s"(ldsa) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) { out[0] = v[0]; out[1] = v[3]; out[2] = lds[0]; out[3] = lds[3]; }
}

int main() {
    uint32_t *buf, *out;
    hipMalloc(&buf, 65536); hipMalloc(&out, 64);
    uint32_t* h = (uint32_t*)malloc(65536);
    for (int i = 0; i < 16384; ++i) h[i] = 0xA5A50000u | (uint32_t)i;
    hipMemcpy(buf, h, 65536, hipMemcpyHostToDevice);
    struct { const char* what; uint32_t voff, soff; } cases[] = {
        {"in range: voffset 0, soffset 0", 0, 0},
        {"in range: voffset 0, soffset 1024", 0, 1024},
        {"voffset past num_records (4096 + 0)", 4096, 0},
        {"voffset in range, voffset + soffset past num_records (0 + 8192)", 0, 8192},
        {"voffset in range, lane 63 crosses with soffset (3072 + 1024)", 3072, 1024},
    };
    for (auto& c : cases) {
        hipMemset(out, 0x11, 64);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, 4096u, c.voff, c.soff, out);
        uint32_t r[4]; hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
        printf("%-72s -> vgpr %08x %08x | lds %08x %08x  (memory there: %08x)\n", c.what, r[0], r[1], r[2], r[3], h[(c.voff + c.soff) / 4]);
    }
    printf("verdict: soffset %s part of the raw-buffer range check on this device\n", "see row 4: 00000000 = IS, a5a5.... = is NOT");
    return 0;
}
