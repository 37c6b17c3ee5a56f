// Internal hooks of libclipa_hip.so for the in-process A/B harnesses (tools/, tools/probes/) and for tests that must prove
// which kernel a shape was dispatched to.  NOT part of the C ABI (include/clipa_hip.h does not declare them, clipa_amd.lib
// does not bind them by default) and inert in production: clipa_internal_debug_set refuses anything but a reset unless the
// process environment holds CLIPA_DEBUG_HOOKS=1.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* Two relaxed process-global atomics read by the GEMM launchers.  gemm_nt_variant: 0 = per-shape default, 1 keeps whole-tile
 * shapes on gemm_nt2 (and whole-tile fp8 shapes on gemm_nt_f8_kernel), 2 + s selects schedule s of gemm_nta.  flags (wrong
 * results where noted): gemm_nt 2 = main loop only, 8 = row-major tile order, bits 20..25 = tile-group size override;
 * gemm_nta / gemm_f8a 64 = epilogue stores dropped by the bounds check, 128 = every tile stores to tile 0; gemm_tn 1024 /
 * 2048 force the 16x16x32 / ping-pong kernel, 4096 / 8192 force the slice-per-XCD / tile-per-XCD work order, 16384 keeps
 * whole-tile shapes off gemm_tna, 32768 selects its schedule 1, 65536 / 131072 = its (and gemm_tn8's) operand-fetch ablations; gemm_tn8
 * bits 26..27 = 1 + schedule of tools/gen_gemm_tn8.py (0 = the default); attention (probe
 * builds with -DCLIPA_ATTN_PERSISTENT_EXPERIMENT only): 262144 / 524288 select the single-sweep backward / persistent forward of
 * tools/probes/attention_persistent/, bits 20 / 21 their timing ablations (no pair arithmetic / no output stores). */
int clipa_internal_debug_set(int gemm_nt_variant, int ablation_flags);
int clipa_internal_debug_flags(void);      /* the flags word, for launchers outside gemm_nt.hip */
/* Which GEMM kernel family the calling process launched last: 0 none, 1 gemm_nt2, 2 gemm_nta, 3 gemm_tn2, 4 gemm_tn3,
 * 5 gemm_tna, 6 gemm_f8a, 7 gemm_nt_f8_kernel. */
int clipa_internal_last_gemm(void);
/* Launches per GEMM kernel family since the last reset: out[1..9] = the families above (8 gemm_tn8, 9 its byte-gather kernel alone),
 * out[10] / out[11] = gemm_nta with the e4m3 pre-activation copy / operand epilogue, out[12] / out[13] = the same of gemm_f8a.
 * Returns the number of slots; reset != 0 zeroes them after the read. */
int clipa_internal_gemm_counts(long* out, int n, int reset);
#ifdef __cplusplus
}
#endif
