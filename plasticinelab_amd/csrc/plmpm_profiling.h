// Profiling-only hooks of the particle kernels: the phase marks of the wave-trace tools.  Nothing here emits code in a normal
// build: the macros expand to nothing unless libplmpm.so is compiled with -DPLB_PHASE_TIMING (profiles/tools/wave_trace_run.py).
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

// The timing-only ablation hooks are not in the product source: profiles/r06_ablate_pad_hooks.patch brings back PLB_ABLATE (no LDS
// atomics / no tile flush / no scatter) and PLB_PAD_VALU (extra FMAs behind the gather) of rounds 3-4, profiles/r04_ablation_hooks.patch
// the replay ablations of round 4 (profiles/r03_notes.md, r04_notes.md say which number came from which).
