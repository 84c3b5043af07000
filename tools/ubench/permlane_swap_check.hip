// What v_permlane32_swap / v_permlane16_swap (gfx950) do to the 16-lane rows of two registers, and the [P0 P2 Q0 Q2] / [P1 P3 Q1 Q3]
// regrouping the 16 x 16 x 32 layer-3 loop of sa_mlp_bf16.hip builds with them.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane_swap_check.hip -o /tmp/pl && /tmp/pl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    const unsigned P = 0x100 + l, Q = 0x200 + l;        // P: rows P0..P3 = lanes 0-15, 16-31, 32-47, 48-63
    u32x2 a = __builtin_amdgcn_permlane32_swap(P, Q, false, false);
    u32x2 b = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false);
    out[l] = a[0]; out[64 + l] = a[1]; out[128 + l] = b[0]; out[192 + l] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap -> P'", "permlane32_swap -> Q'", "then permlane16_swap -> P''", "then permlane16_swap -> Q''"};
    for (int r = 0; r < 4; ++r) {
        printf("%-28s:", names[r]);
        for (int row = 0; row < 4; ++row) { const unsigned v = h[64 * r + 16 * row]; printf("  %c%u", (v >> 8) == 1 ? 'P' : 'Q', ((v & 0xff) >> 4)); }
        printf("\n");
    }
    return 0;
}
