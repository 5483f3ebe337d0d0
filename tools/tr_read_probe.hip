// What does ds_read_b64_tr_b16 (gfx950) return?  LDS is filled with u16 = its own element index; every lane passes the address
// given by one of a few candidate patterns and prints what it got.   hipcc --offload-arch=gfx950 -O2 tools/tr_read_probe.hip -o /tmp/trp && /tmp/trp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef short v4s __attribute__((ext_vector_type(4)));

__global__ void probe(int mode, int rowstride_elems, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr_elems;
    if (mode == 0) addr_elems = 0;                                             // uniform address
    else if (mode == 1) addr_elems = (l & 15) * 4;                             // consecutive 8-byte pieces
    else if (mode == 2) addr_elems = (l & 3) * rowstride_elems + (l >> 2 & 3) * 4 + (l >> 4) * 16;   // 4 rows x 4 pieces per 16 lanes
    else addr_elems = (l & 15) * rowstride_elems + (l >> 4) * 4;              // 16 rows, one 8-byte piece each
    const unsigned byte_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + addr_elems * 2;
    v4s v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)v[j];
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 4; ++mode) {
        const int rs = 64;
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, rs, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (row stride %d elements)\n", mode, rs);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
