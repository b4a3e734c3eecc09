// tools/issue_rate.hip -- what does a wave-instruction cost a gfx950 SIMD?  
//
// k_step is bound by instructions issued.  Rounds 3 - 4 priced that as "4 cycles per wave-instruction of any
// kind", from the saturated 32768-env run.  The guide (MI355X_MICROARCH.md:52-54) says a wave64 VALU instruction occupies the
// SIMD-32 for 2 cycles and that SALU / VALU / LDS / VMEM are separate pipes: if the scalar instructions of one wave issue beside
// the vector instructions of another, the bound of a step is set by max(VALU, SALU) and not by their sum.  This program measures
// it: W = 1 / 2 / 4 / 8 waves per SIMD on every CU, each wave running a loop of one instruction mix, timed with s_memtime inside
// the wave (shader cycles) and with HIP events around the launch.
//
//   hipcc -O2 --offload-arch=gfx950 tools/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate > profiles/r05_issue_rate.txt
//
// Columns: cyc/inst/wave = cycles one wave needs per instruction of its stream; cyc/inst/SIMD = cycles the SIMD spends per
// wave-instruction it retires (= the former / W): the "price" of an instruction in a launch where all W waves are busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Mix {
    VALU_IND, VALU_DEP, SALU_IND, SALU_DEP, MIX_1V1S, MIX_2V1S, MIX_4V1S, SPLIT_VS, LADDER, LADDER_BR, SELECT, VCMP_SEL, LDS_V, TRANS, NMIX
};
static const char* MIXNAME[NMIX] = {
    "valu independent (v_fma_f32, 8 chains)",
    "valu dependent (v_fma_f32, 1 chain)",
    "salu independent (s_add_u32, 8 chains)",
    "salu dependent (s_add_u32, 1 chain)",
    "same wave 1 valu : 1 salu",
    "same wave 2 valu : 1 salu",
    "same wave 4 valu : 1 salu",
    "waves split: half pure valu, half pure salu",
    "ladder: v_cmp + s_and_saveexec + 2 valu + s_or exec (5 insts)",
    "ladder + s_cbranch_execz (not taken) (6 insts)",
    "select: v_cmp + 2 valu + 2 v_cndmask (5 insts)",
    "v_cmp (vcc) + v_cndmask pairs",
    "1 ds_read_b32 : 4 valu",
    "valu transcendental (v_rcp_f32 / v_sqrt_f32, 8 chains)",
};
// instructions per loop body (the loop control -- s_sub + s_cmp + s_cbranch -- is added in insts_per_iter())
#define REP 32

template <int MIX>
__global__ void __launch_bounds__(1024) k_issue(unsigned long long* out, int iters, float seed)
{
    __shared__ float lds[1024];
    lds[threadIdx.x] = seed;
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    unsigned s0 = (unsigned)iters, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    s0 = __builtin_amdgcn_readfirstlane(s0); s1 = __builtin_amdgcn_readfirstlane(s1); s2 = __builtin_amdgcn_readfirstlane(s2);
    s3 = __builtin_amdgcn_readfirstlane(s3); s4 = __builtin_amdgcn_readfirstlane(s4); s5 = __builtin_amdgcn_readfirstlane(s5);
    s6 = __builtin_amdgcn_readfirstlane(s6); s7 = __builtin_amdgcn_readfirstlane(s7);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // waves of a workgroup go to the SIMDs in a cyclic order of period 4: waves w and w + 4 share a SIMD
    const bool second_half = ((wave >> 2) & 1) != 0;
    unsigned lds_addr = (threadIdx.x & 1023) * 4;
    __syncthreads();
    unsigned long long r0 = wall_clock64();  // s_memrealtime: constant 100 MHz
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MIX == VALU_IND || (MIX == SPLIT_VS && !second_half)) {
            asm volatile(".rept 4\n"
                "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        } else if (MIX == VALU_DEP) {
            asm volatile(".rept 32\n v_fma_f32 %0, %0, %0, %0\n .endr" : "+v"(v0));
        } else if (MIX == SALU_IND || (MIX == SPLIT_VS && second_half)) {
            asm volatile(".rept 4\n"
                "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
                "s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
                ".endr" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
        } else if (MIX == SALU_DEP) {
            asm volatile(".rept 32\n s_add_u32 %0, %0, 1\n .endr" : "+s"(s0) : : "scc");
        } else if (MIX == MIX_1V1S) {
            asm volatile(".rept 4\n"
                "v_fma_f32 %0, %0, %0, %0\n s_add_u32 %4, %4, 1\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %5, %5, 1\n"
                "v_fma_f32 %2, %2, %2, %2\n s_add_u32 %6, %6, 1\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %7, %7, 1\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        } else if (MIX == MIX_2V1S) {  // 30 instructions
            asm volatile(".rept 5\n"
                "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %4, %4, 1\n"
                "v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %5, %5, 1\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1) : : "scc");
        } else if (MIX == MIX_4V1S) {  // 30 instructions
            asm volatile(".rept 6\n"
                "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %4, %4, 1\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0) : : "scc");
        } else if (MIX == LADDER) {  // 30 instructions: what a short divergent `if` costs
            asm volatile(".rept 6\n"
                "v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                "s_or_b64 exec, exec, s[20:21]\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc", "scc", "s20", "s21");
        } else if (MIX == LADDER_BR) {  // 30 instructions: the same with the skip branch the compiler adds (never taken here)
            asm volatile(".rept 5\n"
                "v_cmp_lt_f32 vcc, %0, %1\n s_and_saveexec_b64 s[20:21], vcc\n s_cbranch_execz 1f\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                "1:\n s_or_b64 exec, exec, s[20:21]\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc", "scc", "s20", "s21");
        } else if (MIX == SELECT) {  // 30 instructions: the branch-free form of the same `if`
            asm volatile(".rept 6\n"
                "v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 %4, %2, %2, %2\n v_fma_f32 %5, %3, %3, %3\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %5, vcc\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : : "vcc");
        } else if (MIX == VCMP_SEL) {  // 32 instructions
            asm volatile(".rept 16\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n .endr"
                : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : : "vcc");
        } else if (MIX == LDS_V) {  // 30 instructions, the reads waited for at the end of the body
            asm volatile(".rept 6\n"
                "ds_read_b32 %4, %5\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                ".endr\n s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "=v"(v4) : "v"(lds_addr));
        } else if (MIX == TRANS) {
            asm volatile(".rept 4\n"
                "v_rcp_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_rcp_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                "v_rcp_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_rcp_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                ".endr" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long r1 = wall_clock64();
    float keep = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + (float)(s0 + s1 + s2 + s3 + s4 + s5 + s6 + s7);
    if (keep == 123.456f) lds[0] = keep;  // keeps the chains alive
    if ((threadIdx.x & 63) == 0) {
        out[((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0;
        out[((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = r1 - r0;
    }
    if (lds[0] == -1.0f) out[0] = 0;
}

static int body_insts(int mix) {
    switch (mix) {
        case MIX_2V1S: case MIX_4V1S: case LADDER: case LADDER_BR: case SELECT: case LDS_V: return 30 + (mix == LDS_V ? 1 : 0);
        default: return 32;
    }
}

typedef void (*kern_t)(unsigned long long*, int, float);
static kern_t KERN[NMIX] = {
    k_issue<VALU_IND>, k_issue<VALU_DEP>, k_issue<SALU_IND>, k_issue<SALU_DEP>, k_issue<MIX_1V1S>, k_issue<MIX_2V1S>, k_issue<MIX_4V1S>,
    k_issue<SPLIT_VS>, k_issue<LADDER>, k_issue<LADDER_BR>, k_issue<SELECT>, k_issue<VCMP_SEL>, k_issue<LDS_V>, k_issue<TRANS>,
};

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    int sclk_khz = 0;
    CK(hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0));
    printf("# %s, %d CUs, shader clock (attribute) %.0f MHz; loop body %d iterations\n", p.gcnArchName, cus, sclk_khz / 1e3, iters);
    printf("# cyc = s_memtime ticks inside the wave (shader cycles, MI355X_MICROARCH.md:446), averaged over all waves of the launch;\n");
    printf("# ns = the launch's wall time (HIP events, best of 3) / instructions per wave -- includes the ~3 us launch ramp.\n");
    unsigned long long* d;
    CK(hipMalloc(&d, sizeof(unsigned long long) * cus * 4 * 32 * 2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("# clk MHz = s_memtime ticks / s_memrealtime (100 MHz) over the wave's life; life/launch = the share of the launch the average wave lived.\n");
    printf("%-62s %3s %10s %13s %13s %13s %13s %9s %9s\n", "mix", "W", "launch us", "cyc/inst/wave", "cyc/inst/SIMD", "ns/inst/wave", "ns/inst/SIMD",
           "clk MHz", "life/lnch");
    std::vector<unsigned long long> h((size_t)cus * 4 * 32 * 2);
    for (int mix = 0; mix < NMIX; ++mix) {
        for (int W : {1, 2, 4, 8}) {
            if (mix == SPLIT_VS && W < 2) continue;
            // W waves per SIMD on every CU: blocks of min(W, 4) * 256 threads, (W > 4 ? 2 : 1) blocks per CU
            int threads = std::min(W, 4) * 256;
            int blocks = cus * (W > 4 ? W / 4 : 1);
            if (mix == SPLIT_VS && W == 8) { /* blocks of 16 waves: waves 0-3 valu, 4-7 salu, 8-11 valu, 12-15 salu */ }
            int total_waves = blocks * threads / 64;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(KERN[mix], dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f + rep);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) best = std::min(best, ms);
            }
            CK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * total_waves * 2, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < total_waves; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
            cyc /= total_waves;
            real /= total_waves;  // ticks of 10 ns the average wave lived
            const double mhz = real > 0 ? cyc / (real * 0.01) : 0;  // s_memtime ticks per microsecond of wave life
            double insts = (double)iters * (body_insts(mix) + 3);  // + s_add / s_cmp / s_cbranch of the loop
            double ns_wave = best * 1e6 / insts;
            printf("%-62s %3d %10.1f %13.3f %13.3f %13.3f %13.3f %9.0f %9.2f\n", MIXNAME[mix], W, best * 1e3, cyc / insts, cyc / insts / W,
                   ns_wave, ns_wave / W, mhz, real * 0.01 / (best * 1e3));
        }
    }
    CK(hipFree(d));
    return 0;
}
