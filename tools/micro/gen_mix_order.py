# usage: python3 gen_mix_order.py  (writes mix_order.hip in the current directory; build it like the other microbenchmarks)
# generates mix_order.hip: ordering effects between 'fast' (v_add_u32) and 'slow' (v_mad_i32_i24) instructions
pats = {}
def rep(unit, n=64):
    s = (unit * (n // len(unit) + 1))[:n]
    return s
pats['F1S1'] = rep('FS'); pats['F2S2'] = rep('FFSS'); pats['F4S4'] = rep('F'*4+'S'*4)
pats['F8S8'] = rep('F'*8+'S'*8); pats['F16S16'] = rep('F'*16+'S'*16); pats['F32S32'] = 'F'*32+'S'*32
pats['F2S1'] = rep('FFS', 63); pats['F8S4'] = rep('F'*8+'S'*4, 60); pats['F32S16'] = rep('F'*32+'S'*16, 48) 
pats['F40S20'] = 'F'*40+'S'*20
pats['allF'] = 'F'*64; pats['allS'] = 'S'*64
out = ['#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>\n']
out.append('''template <int P, int DEP, int SKEW>
__global__ __launch_bounds__(1024) void k(uint64_t* out, int iters, uint32_t seed) {
    uint32_t a[8], b = seed ^ (threadIdx.x & 63), c = seed * 3 + 7;
    for (int i = 0; i < 8; ++i) a[i] = seed + i * 77 + threadIdx.x;
    if (SKEW) { const int w = threadIdx.x >> 6; for (int i = 0; i < (w >> 2) * 7 + 3; ++i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c)); }
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
''')
names = list(pats)
for pi, nm in enumerate(names):
    out.append('        if constexpr (P == %d) {\n' % pi)
    for j, ch in enumerate(pats[nm]):
        # DEP: all on one accumulator (dependent chain); else rotate 8
        idx = '(DEP ? 0 : %d)' % (j % 8)
        if ch == 'F':
            out.append('            asm volatile("v_add_u32 %%0, %%0, %%1" : "+v"(a[%s]) : "v"(b));\n' % idx)
        else:
            out.append('            asm volatile("v_mad_i32_i24 %%0, %%0, %%1, %%2" : "+v"(a[%s]) : "v"(b), "v"(c));\n' % idx)
    out.append('        }\n')
out.append('''    }
    const uint64_t t1 = __builtin_readcyclecounter();
    uint32_t s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 0x12345678u) out[1] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[2 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int P, int DEP, int SKEW>
void run(const char* name, int n, int nf, uint64_t* d_out) {
    const int iters = 2000;
    double res[4];
    for (int wps = 1; wps <= 4; ++wps) {
        hipLaunchKernelGGL((k<P, DEP, SKEW>), dim3(256), dim3(256 * wps), 0, 0, d_out, iters, 12345u);
        hipLaunchKernelGGL((k<P, DEP, SKEW>), dim3(256), dim3(256 * wps), 0, 0, d_out, iters, 12345u);
        hipDeviceSynchronize();
        uint64_t h[32];
        hipMemcpy(h, d_out, 32 * 8, hipMemcpyDeviceToHost);
        uint64_t mx = 0;
        for (int i = 0; i < 4 * wps; ++i) mx = h[2 + i] > mx ? h[2 + i] : mx;
        res[wps - 1] = (double)mx / (iters * (double)n) / wps;
    }
    printf("%-10s dep=%d skew=%d fast=%.2f  | 1w %.2f  2w %.2f  3w %.2f  4w %.2f  (ideal %.2f)\\n", name, DEP, SKEW, (double)nf / n, res[0], res[1], res[2], res[3],
           (nf * 2.4 + (n - nf) * 4.25) / n);
}
int main() {
    uint64_t* d_out; hipMalloc(&d_out, 512); hipMemset(d_out, 0, 512);
''')
for pi, nm in enumerate(names):
    n = len(pats[nm]); nf = pats[nm].count('F')
    for dep, skew in ((0, 0), (0, 1), (1, 0)):
        out.append('    run<%d, %d, %d>("%s", %d, %d, d_out);\n' % (pi, dep, skew, nm, n, nf))
out.append('    return 0;\n}\n')
open('mix_order.hip', 'w').write(''.join(out))
