// valu_issue.hip -- what does ONE wave64 instruction cost a gfx950 SIMD?  ("settle what bounds
// the adjoint tracer": rocprof's SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 1.0 quad-cycle per VALU instruction, which is
// 91 % busy at 4 cycles per instruction and 36 % at the 2 cycles the SIMD-32 datapath suggests.)
//
// Every kernel runs `iters` x 8 independent instructions of one kind per wave (8 accumulators, inline asm, no
// memory traffic) at w = 1, 2, 4, 8 waves per SIMD (256 CUs x w workgroups of 256 threads) and reports
//   cycles per wave-instruction per SIMD = elapsed shader cycles (s_memtime, slowest wave) / (w * iters * 8).
// At w = 1 that is the dependent-issue cost of the wave; at w = 8 the throughput cost the SIMD pays.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define A8(OPSTR) \
    asm volatile(OPSTR : "+v"(a0) : "v"(b)); asm volatile(OPSTR : "+v"(a1) : "v"(b)); \
    asm volatile(OPSTR : "+v"(a2) : "v"(b)); asm volatile(OPSTR : "+v"(a3) : "v"(b)); \
    asm volatile(OPSTR : "+v"(a4) : "v"(b)); asm volatile(OPSTR : "+v"(a5) : "v"(b)); \
    asm volatile(OPSTR : "+v"(a6) : "v"(b)); asm volatile(OPSTR : "+v"(a7) : "v"(b));

#define A8_VCC(OPSTR) \
    asm volatile(OPSTR : "+v"(a0) : "v"(b) : "vcc"); asm volatile(OPSTR : "+v"(a1) : "v"(b) : "vcc"); \
    asm volatile(OPSTR : "+v"(a2) : "v"(b) : "vcc"); asm volatile(OPSTR : "+v"(a3) : "v"(b) : "vcc"); \
    asm volatile(OPSTR : "+v"(a4) : "v"(b) : "vcc"); asm volatile(OPSTR : "+v"(a5) : "v"(b) : "vcc"); \
    asm volatile(OPSTR : "+v"(a6) : "v"(b) : "vcc"); asm volatile(OPSTR : "+v"(a7) : "v"(b) : "vcc");

template <int OP>
__global__ void __launch_bounds__(256) issue_kernel(uint32_t *sink, int iters, unsigned long long *cycles)
{
    uint32_t a0 = threadIdx.x + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = (threadIdx.x * 2654435761u) | 1u;
    unsigned long long q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
    const unsigned long long smask = 0x5555aaaa3333ccccull ^ (unsigned long long) iters;
    uint32_t acc_s = 0;
    asm volatile("s_mov_b64 vcc, %0" :: "s"(smask) : "vcc");
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (OP == 0) { A8("v_fma_f32 %0, %0, %1, %0") }
        else if constexpr (OP == 1) { A8("v_add_u32 %0, %0, %1") }
        else if constexpr (OP == 2) { A8("v_mul_lo_u32 %0, %0, %1") }
        else if constexpr (OP == 3) { A8("v_mul_hi_u32 %0, %0, %1") }
        else if constexpr (OP == 4) {
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q0) : "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q1) : "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q2) : "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q3) : "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q4) : "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q5) : "v"(b) : "vcc");
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q6) : "v"(b) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q7) : "v"(b) : "vcc");
        }
        else if constexpr (OP == 5) { A8_VCC("v_cndmask_b32 %0, %0, %1, vcc") }
        else if constexpr (OP == 6) { A8("v_log_f32 %0, %0") }
        else if constexpr (OP == 7) { A8("v_rcp_f32 %0, %0") }
        else if constexpr (OP == 8) { A8("ds_bpermute_b32 %0, %1, %0") asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if constexpr (OP == 9) {
            asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q0)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q1));
            asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q2)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q3));
            asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q4)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q5));
            asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q6)); asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(q7));
        }
        else if constexpr (OP == 10) { A8("v_xor_b32 %0, %0, %1") }
        else if constexpr (OP == 11) { A8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") }
        else if constexpr (OP == 12) { A8("v_sqrt_f32 %0, %0") }
        else if constexpr (OP == 13) { A8_VCC("v_cmp_lt_f32 vcc, %0, %1") }
        else if constexpr (OP == 14) { A8("v_floor_f32 %0, %0") }
        else if constexpr (OP == 15) { A8("v_cvt_i32_f32 %0, %0") }
        else if constexpr (OP == 16) { A8("v_alignbit_b32 %0, %0, %0, %1") }
        else if constexpr (OP == 17) {   // mixed: one s_* between v_* (SALU co-issue)
            A8("v_fma_f32 %0, %0, %1, %0") asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        }
        else if constexpr (OP == 18) { A8("v_mul_f32 %0, %0, %1") }
        else if constexpr (OP == 20) { A8("v_cndmask_b32 %0, %0, %1, vcc") }              // vcc set once before the loop, never clobbered
        else if constexpr (OP == 21) {
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "s"(smask)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "s"(smask));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "s"(smask)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "s"(smask));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "s"(smask)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "s"(smask));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "s"(smask)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "s"(smask));
        }
        else if constexpr (OP == 22) {                                                       // v_cmp -> v_cndmask pairs (what a select compiles to)
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b) : "vcc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b) : "vcc"); asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b) : "vcc");
        }
        else if constexpr (OP == 30) {                                                       // ONE v_cmp -> vcc, EIGHT v_cndmask reading it
            asm volatile("v_cmp_lt_u32 vcc, %8, %9\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n v_cndmask_b32 %3, %3, %9, vcc\n"
                         "v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");
        }
        else if constexpr (OP == 31) {                                                       // s_and_b64 vcc (SALU-written), then 8 v_cndmask
            asm volatile("s_and_b64 vcc, %9, exec\n s_nop 1\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(smask) : "vcc");
        }
        else if constexpr (OP == 32) {                                                       // v_cmp -> vcc, 4 unrelated VALU, then cndmask (x2 per iteration)
            asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %1, %1, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_xor_b32 %2, %2, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
            asm volatile("v_cmp_lt_u32 vcc, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %1, %1, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_xor_b32 %2, %2, %4"
                         : "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
        }                                                                                    // (14 instructions; reported per 8)
        else if constexpr (OP == 33) {                                                       // e64 encoding with VCC as the mask pair
            A8("v_cndmask_b32_e64 %0, %0, %1, vcc")
        }
        else if constexpr (OP == 34) {                                                       // v_cmp_e64 -> SGPR pair, 8 v_cndmask_e64 reading it
            unsigned long long m;
            asm volatile("v_cmp_lt_u32_e64 %0, %1, %2" : "=s"(m) : "v"(a0), "v"(b));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "s"(m));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "s"(m));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a4) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a5) : "v"(b), "s"(m));
            asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a6) : "v"(b), "s"(m)); asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a7) : "v"(b), "s"(m));
        }                                                                                    // (9 instructions; reported per 8)
        else if constexpr (OP == 35) {                                                       // exec-masked VALU: s_and_saveexec + 8 v_mov + restore
            unsigned long long sv;
            asm volatile("s_and_saveexec_b64 %0, %1" : "=s"(sv) : "s"(smask) : "exec");
            A8("v_xor_b32 %0, %0, %1")
            asm volatile("s_mov_b64 exec, %0" :: "s"(sv) : "exec");
        }
        else if constexpr (OP == 23) {                                                       // v_readlane_b32 into SGPRs
            uint32_t s0, s1, s2, s3;
            asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            asm volatile("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %5, 5\n v_readlane_b32 %2, %6, 7\n v_readlane_b32 %3, %7, 9"
                         : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));
            acc_s += s0 ^ s1 ^ s2 ^ s3;
        }
        else if constexpr (OP == 24) { A8("v_mul_u32_u24 %0, %0, %1") }
        else if constexpr (OP == 25) { A8("v_mad_u32_u24 %0, %0, %1, %0") }
        else if constexpr (OP == 26) { A8("v_max_f32 %0, %0, %1") }
        else if constexpr (OP == 27) { A8("v_min_i32 %0, %0, %1") }
        else if constexpr (OP == 28) { A8("v_lshl_add_u32 %0, %0, 2, %1") }
        else if constexpr (OP == 29) { A8("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"00p11\")") asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        else if constexpr (OP == 19) {
            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a0) : "v"(b) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a1) : "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a2) : "v"(b) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a3) : "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a4) : "v"(b) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a5) : "v"(b) : "vcc");
            asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a6) : "v"(b) : "vcc"); asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a7) : "v"(b) : "vcc");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = acc_s ^ a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t) (q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7);
    if (r == 0x12345678u) sink[0] = r;
    if ((threadIdx.x & 63) == 0) atomicMax(cycles, t1 - t0);
}

template <int OP>
int run(const char *name, uint32_t *sink, unsigned long long *d_cycles, int n_cus)
{
    const int iters = 4096;
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        double best = 1e30, best_ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(d_cycles, 0, sizeof(unsigned long long)));
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(issue_kernel<OP>, dim3(n_cus * w), dim3(256), 0, 0, sink, iters, d_cycles);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            unsigned long long c = 0;
            CHECK(hipMemcpy(&c, d_cycles, sizeof c, hipMemcpyDeviceToHost));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double per = (double) c / ((double) w * iters * 8);
            if (per < best) { best = per; best_ms = ms; }
        }
        // wall-clock view: instructions per SIMD / time -> cycles at 2.4 GHz
        const double wall_cyc = best_ms * 1e-3 * 2.4e9 / ((double) w * iters * 8);
        printf("  w=%d: %6.2f (wall@2.4GHz %5.2f)", w, best, wall_cyc);
        (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    }
    printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz; columns: s_memtime-cycles per wave64 instruction per SIMD at w waves/SIMD\n",
           prop.name, n_cus, prop.clockRate);
    uint32_t *sink; unsigned long long *d_cycles;
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&d_cycles, 8));
    run<0>("v_fma_f32", sink, d_cycles, n_cus);
    run<18>("v_mul_f32", sink, d_cycles, n_cus);
    run<1>("v_add_u32", sink, d_cycles, n_cus);
    run<10>("v_xor_b32", sink, d_cycles, n_cus);
    run<5>("v_cndmask_b32 (s_nop between)", sink, d_cycles, n_cus);
    run<20>("v_cndmask_b32 e32 vcc", sink, d_cycles, n_cus);
    run<21>("v_cndmask_b32_e64 sgpr mask", sink, d_cycles, n_cus);
    run<22>("v_cmp+v_cndmask pair x4 (per 8)", sink, d_cycles, n_cus);
    run<30>("1 v_cmp + 8 v_cndmask vcc (per 8)", sink, d_cycles, n_cus);
    run<31>("s_and vcc + 8 v_cndmask vcc (per 8)", sink, d_cycles, n_cus);
    run<32>("cmp, 4 valu, cndmask x2 (14 per 8)", sink, d_cycles, n_cus);
    run<33>("v_cndmask_b32_e64 ... vcc", sink, d_cycles, n_cus);
    run<34>("v_cmp_e64 sgpr + 8 cndmask_e64", sink, d_cycles, n_cus);
    run<35>("saveexec + 8 v_xor + restore", sink, d_cycles, n_cus);
    run<23>("v_readlane_b32 x8", sink, d_cycles, n_cus);
    run<24>("v_mul_u32_u24", sink, d_cycles, n_cus);
    run<25>("v_mad_u32_u24", sink, d_cycles, n_cus);
    run<26>("v_max_f32", sink, d_cycles, n_cus);
    run<27>("v_min_i32", sink, d_cycles, n_cus);
    run<28>("v_lshl_add_u32", sink, d_cycles, n_cus);
    run<29>("ds_swizzle_b32 bitmask (8/waitcnt)", sink, d_cycles, n_cus);
    run<13>("v_cmp_lt_f32 vcc", sink, d_cycles, n_cus);
    run<19>("v_add_co/addc_co pair (per instr)", sink, d_cycles, n_cus);
    run<16>("v_alignbit_b32", sink, d_cycles, n_cus);
    run<14>("v_floor_f32", sink, d_cycles, n_cus);
    run<15>("v_cvt_i32_f32", sink, d_cycles, n_cus);
    run<2>("v_mul_lo_u32", sink, d_cycles, n_cus);
    run<3>("v_mul_hi_u32", sink, d_cycles, n_cus);
    run<4>("v_mad_u64_u32", sink, d_cycles, n_cus);
    run<9>("v_lshlrev_b64", sink, d_cycles, n_cus);
    run<6>("v_log_f32", sink, d_cycles, n_cus);
    run<7>("v_rcp_f32", sink, d_cycles, n_cus);
    run<12>("v_sqrt_f32", sink, d_cycles, n_cus);
    run<11>("v_mov_b32_dpp row_shr:1", sink, d_cycles, n_cus);
    run<8>("ds_bpermute_b32 (8 per waitcnt)", sink, d_cycles, n_cus);
    run<17>("8 v_fma_f32 + 4 s_nop (per v_fma)", sink, d_cycles, n_cus);
    return 0;
}
