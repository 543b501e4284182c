/* katamx_tuning.h - kernel-tuning instrumentation exported by libkatamx.so beside the C ABI of include/katamx.h.
 *
 * NOT part of the drop-in boundary: nothing a KataGo binding needs is declared here. These entry points time single kernels on
 * synthetic data, expose the convolution launcher's shape choice to the tests, and measure the matrix cores' issue rate; they are
 * called by tools/*.py, bench.py's `box` field and tests/test_conv_chooser.py through katago_amd/capi.py (TUNING_SIGNATURES).
 * (They sat in include/katamx.h until ABI 6.) */
#ifndef KATAMX_TUNING_H_
#define KATAMX_TUNING_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Average duration (ms, hipEvents) of one launch of the bf16 convolution kernel on synthetic data: kernel size ks,
 * wn = 32-channel tiles per wave, variant = 0 (product kernel) or depth*1000 + ablation mask (conv_kernel.h),
 * epilogue_mode 0 = BN+act output, 1 = residual + raw + BN+act outputs. Kernel tuning instrumentation. */
int kmx_bench_conv(int ks, int wn, int variant, int cin, int cout, int batch, int nn_x_len, int nn_y_len,
                   int epilogue_mode, int iters, double* avg_ms);
/* The same layer on n_streams streams at once (each its own `batch` boards, `launches` back-to-back launches), stream i
 * started i * delay_us microseconds after stream 0: what a stagger of a FRACTION of a launch between co-resident
 * work-groups buys. total_ms = wall time from the common start to the last stream's end. Kernel tuning instrumentation. */
int kmx_bench_conv_streams(int ks, int cfg, int cin, int cout, int batch, int n_streams, double delay_us, int launches,
                           int epilogue_mode, double* total_ms);
/* Average duration (ms) of one launch of the fused seam kernel (192 -> 384 -> 192, mish, bf16) on `batch` 19x19 boards of
 * synthetic data; timing != 0 runs the instrumented instantiation of the persistent kernel and prints per-wave cycle sums of
 * its phases on stderr. KMX_PW_V2=0 selects the one-tile-per-work-group kernel. Kernel tuning instrumentation. */
int kmx_bench_seam(int batch, int iters, int timing, double* avg_ms);
/* n_conv (2 | 4) convolutions 3x3 192 -> 192 on `batch` 19x19 boards, per sequence: chained = 0 one launch each, 2 | 4 chained launches
 * (conv_chain_kernel.h); timing != 0: the chained launches print cycle stamps per phase to stderr. KMX_BENCH_DTYPE=fp16 | bf16 (default). */
int kmx_bench_conv_chain(int batch, int n_conv, int chained, int iters, int timing, double* avg_ms);
/* Host-only introspection of the convolution launcher (no device needed): the work-group shape chosen for a kernel size,
 * a padded channel count (multiple of 64) and a batch, and whether a kernel of that shape exists and tiles the channels.
 * tests/test_conv_chooser.py walks every combination the engine can ask for. */
int kmx_debug_conv_cfg(int ks, int cout_pad, int batch, int* cfg, int* instantiated);

/* MFMA issue-rate microbenchmark (v_mfma_f32_32x32x16_bf16, 18 per step as in the convolution): mode bit 1 adds an
 * s_barrier per step, bit 2 adds the step's 12 ds_read_b128. Reports the rate and the shader clock it ran at: the
 * practical ceiling the convolution is measured against. Kernel tuning instrumentation. */
int kmx_bench_mfma(int waves_per_wg, int wgs, int mode, int steps, int iters, double* avg_ms, double* tflops, double* core_mhz);

/* What the matrix cores SUSTAIN, by operand data (round 6): the loop above - shape bit 0 clear: the bare chain of 18 MFMAs per step on a 3 x 3
 * tile of accumulators, set: the convolution's step (+ 12 ds_read_b128 + one s_barrier); shape bits 1-2 (fp16 only): the order in which a k
 * half's nine MFMAs go out - 0 weight fragment outer (the convolution's), 1 the same as a snake, 2 image fragment outer - on `wgs` work-groups of 8 waves for `seconds` seconds,
 * operands of data_kind 0 zeros, 1 a smooth ramp of small positive numbers (what kmx_bench_mfma multiplies), 2 uniform noise in [-1, 1),
 * 3 the distributions of the bench's own operands (normal weights of a random-init 192-channel 3x3 layer x mish of a unit normal at 1/8);
 * precision_mode KMX_PREC_FP16 | KMX_PREC_BF16. TFLOP/s over the whole run and the shader clock inside the last launch: on noise the chip
 * clocks down whatever else the kernel does, and that - not 2.4 GHz x 256 CUs - is the rate a convolution on real activations can be held
 * against. tools/mfma_power_probe.py, bench.py's `box`. */
int kmx_bench_mfma_sustained(int wgs, int shape, int data_kind, int precision_mode, double seconds, double* tflops, double* core_mhz);

/* What a dependent launch costs before it does anything: `launches` dependent launches of a 512-thread kernel with lds_bytes of LDS on
 * `wgs` work-groups, captured in one hipGraph and replayed `iters` times; us per launch. mode 0: the kernel ends at once; 1: every lane
 * loads 16 bytes the launch before stored and stores them again; 2: as 1 through LDS and a barrier. tools/launch_floor.py. */
int kmx_bench_launch_floor(int wgs, int lds_bytes, int mode, int launches, int iters, double* us_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* KATAMX_TUNING_H_ */
