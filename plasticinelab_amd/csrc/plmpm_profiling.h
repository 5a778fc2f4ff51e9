// Profiling-only hooks of the particle kernels.  Nothing here emits code in a normal build: the macros expand to
// nothing (or to a constant-false test) unless libplmpm.so is compiled with -DPLB_PHASE_TIMING / -DPLB_ABLATE=...
// (profiles/r0N_notes.md name the builds that were measured this way; the Makefile's EXTRA carries the flags).
#pragma once

// profiling builds only (-DPLB_PHASE_TIMING): PT_MARK(k) stamps s_memtime at the end of phase k; for the launch of
// frame PLB_TRACE_FRAME every wave stores its stamps and its hardware id (XCC / SE / CU / SIMD) with plain stores to
// D.trace[(kernel slot) * 16384 * 16 + wave * 16 + ...] (plmpm_debug_trace).  No atomics: same-address atomics from
// every wave clog the memory pipeline and become the thing being measured.  Nothing is emitted in normal builds.
#ifdef PLB_PHASE_TIMING
#ifndef PLB_TRACE_FRAME
#define PLB_TRACE_FRAME 20
#endif
#define PT_BEGIN() unsigned long long pt_abs[11] = {__builtin_readcyclecounter(), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PT_MARK(k) do { pt_abs[1 + (k)] = __builtin_readcyclecounter(); } while (0)
#define PT_END(D, slot0) do { if ((threadIdx.x & 63) == 0 && f == PLB_TRACE_FRAME) { \
            unsigned long long* q_ = D.trace + ((size_t)((slot0) / 10) * 16384 + blockIdx.x * 4 + (threadIdx.x >> 6)) * 16; \
            for (int k_ = 0; k_ < 11; ++k_) q_[k_] = pt_abs[k_]; \
            q_[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
            q_[12] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); } } while (0)
#else
#define PT_BEGIN() do {} while (0)
#define PT_MARK(k) do {} while (0)
#define PT_END(D, slot0) do {} while (0)
#endif

// Ablations (timing only -- results are wrong): PLB_ABLATE bits 1 no LDS atomics, 2 no tile flush, 4 no scatter at all.
#ifndef PLB_ABLATE
#define PLB_ABLATE 0
#endif
// inside a scatter lambda: stop here when ablation bit `bit` is set, keeping `sum` alive so the arithmetic before it is not
// optimised away
#define PLB_ABLATE_STOP(bit, sum, tile) do { if (PLB_ABLATE & (bit)) { if ((sum) == T(-1e30)) (tile)[0].x = 1.0; return; } } while (0)

// Experiment (profiles/r03_notes.md): PLB_PAD_VALU=N inserts N extra fp32 FMAs (four independent chains) per lane into
// k_g2p_p2g behind its gather -- does the kernel's time follow its VALU instruction count?
#ifndef PLB_PAD_VALU
#define PLB_PAD_VALU 0
#endif
#define PLB_PAD(seed, errp)                                                                    \
    do {                                                                                       \
        if (PLB_PAD_VALU > 0) {                                                                \
            float a0_ = (float)(seed), a1_ = a0_ + 1.f, a2_ = a0_ + 2.f, a3_ = a0_ + 3.f;      \
            _Pragma("unroll") for (int i_ = 0; i_ < PLB_PAD_VALU / 4; ++i_) {                  \
                a0_ = __builtin_fmaf(a0_, 1.0001f, 0.5f); a1_ = __builtin_fmaf(a1_, 1.0002f, 0.25f);   \
                a2_ = __builtin_fmaf(a2_, 0.9999f, 0.125f); a3_ = __builtin_fmaf(a3_, 0.9998f, 0.75f); \
            }                                                                                  \
            if (a0_ + a1_ + a2_ + a3_ == 123456.789f) atomicOr((errp), 64);                    \
        }                                                                                      \
    } while (0)


// The timing-only hooks of round 4's replay ablations (PLB_ABL_PACK / NOSORT / FUSEBWD / ST4 / CONST, PLB_EXP_DIRECT / NOPOSE, PLB_STAGGER,
// the reversed / interleaved dispatch orders and the LDS-DMA prefetch of k_p2g_grad) are not in the product source: apply
// profiles/r04_ablation_hooks.patch to get them back (profiles/r04_notes.md says which number came from which).
