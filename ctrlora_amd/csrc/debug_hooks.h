// Probe hooks of libctrlora_hip.so: A/B switches between schedules / launch forms that compute the SAME result.
// They are exported for tests/tools/attn_bench.py, the A/B environment switches of ctrlora_amd/hip.py and the GPU
// tests that cover every form that ships; they are deliberately NOT declared in include/ctrlora_hip.h (the drop-in
// boundary): nothing a caller of the library needs, no effect on the input contract of any entry point.
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
/* attention schedule: 0 = default; 1 = tile-synchronous kernels only; 11 = backward with s_setprio;
 * 13 / 14 = hybrid forward (fragment lookahead 3 / 2) also where the pre-scaled-Q forward (attention_fwd40.hip) would
 * apply.  Unknown codes: CL_EINVAL, nothing changes. */
int cl_debug_attention_variant(int variant);   /* also 21 = 0 with the dK/dV kernel at FOUR key fragments per wave (round 6 probe: slower) */
/* 1 (default) = the dQ kernel forms delta itself; 0 = separate attn_delta launch */
int cl_debug_attention_fuse_delta(int on);
/* GroupNorm launch forms: three_pass = 1 forces partial -> finalize -> apply; one_pass = 0 disables the one-launch
 * register-resident form (defaults 0, 1) */
int cl_debug_groupnorm_form(int three_pass, int one_pass);
/* 1 = GroupNorms whose groups span >= 1024 pixels run as one cooperative launch (csrc/norm_coop.hip: pixel slabs in
 * registers, the workgroups of a sample meet at a counter); 0 (default: measured no faster, see norm_coop.hip) = the forms above only.  _timeouts: how many workgroups ever gave
 * up waiting at that counter (0 unless something is broken; a timed-out launch produced wrong numbers) */
int cl_debug_groupnorm_coop(int on);
int cl_debug_groupnorm_coop_timeouts(void);
/* Launch tags for the contraction kernels (profiling aid: which SHAPE is a gemm dispatch of a kernel trace?).  While on, every
 * cl_gemm product signature {dtype, mode, M, N, K1, K2, act, residual} gets a small integer tag in order of first appearance
 * (1 .. 255) and each kernel it launches gets `tag` extra workgroups that exit at once -- so a trace row's Grid_Size names the
 * signature: workgroups = real grid + tag (tools/prof_shapes.py decodes it from the table below).  Results are unchanged.
 * cl_debug_gemm_tag_get(i, out): out[0..11] = dtype, mode, M, N, K1, K2, act, has_residual, tag, real workgroups of the main
 * kernel, workgroup size, launches so far; returns CL_EINVAL past the end. */
int cl_debug_gemm_tag(int on);
/* 1 (default) = product signatures WITHOUT a launch-table entry take the x-stationary kernel (gemm_xs.hip) where the rule in
 * gemm.hip says so; 0 = only where a table entry names configuration 34 (CTRLORA_GEMM_XS=0 sets 0 and drops those entries too) */
int cl_debug_gemm_xs_rules(int on);
/* LDS ring depth of the weight-gradient kernel: 3 (default since round 5: three workgroups per CU), 4 or 6 */
int cl_debug_wgrad_ring(int slots);
int cl_debug_gemm_tag_count(void);
int cl_debug_gemm_tag_get(int i, long* out12);
#ifdef __cplusplus
}
#endif
