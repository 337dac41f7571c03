/* imvoxel_lab.h -- measurement and A/B entry points of libimvoxel_hip.so.  NOT part of the operator ABI a reference maintainer binds
 * (include/imvoxel.h): per-thread knobs that force a kernel variant for the A/B tools under tools/, and the two micro-benchmarks bench.py
 * prices its roofline fractions against.  Defaults (never calling anything here) are the product's behaviour. */
#ifndef IMVOXEL_LAB_H_
#define IMVOXEL_LAB_H_
#include "imvoxel.h"
#ifdef __cplusplus
extern "C" {
#endif

/* A/B knob (per calling thread) of the Winograd-domain GEMMs on fp16 pairs: -1 (default) = the z-halo kernel where it applies (1x1x3 along z,
 * stride 1, pad 1, Cin % 32 == 0: one staged tile serves the three z-taps), 0 = the generic LDS-DMA kernel always, 1 .. 4 = force a config. */
int ivx_conv_set_halo_mode(int mode);

/* Tuning knob for A/B experiments only (per calling thread): 0 = automatic tile choice (default); 1..7 force a tile
 * of the generic kernel, 41..53 of the LDS-DMA fp32 kernel, 61..73 of its bf16 instantiation. */
int ivx_conv_set_tile_override(int cfg);
/* Per calling thread, A/B only (tools/wino_ab.py): kernel of the F(6x6,3x3) output transform.  -1 = the library's rule (2 with a
 * residual, 1 without); 0 whole 8x8 tile per thread, 2 channels per lane (the round-2 kernel); 1 the same with 1 channel per lane;
 * 2 buffer addressing + column accumulation, 2 channels per lane; 3 the same with 1 channel per lane.  input_variant: -1 / 0 two
 * channels per lane, 1 one channel per lane. */
int ivx_conv_winograd_set_variant(int32_t output_variant, int32_t input_variant);
/* Per calling thread, A/B only: 1 = one-channel-per-lane epilogue stores in the LDS-DMA conv kernel; 0 (default) = the
 * LDS-transposed epilogue (a lane stores 4 consecutive channels as one 16-byte word) wherever it applies. */
int ivx_conv_set_epilogue_mode(int narrow);
/* Per calling thread, A/B only: 1 = the round-1 tile rule of the direct convolution planner, 0 (default) = scored choice. */
int ivx_conv_set_plan_mode(int mode);
/* Per calling thread, A/B and tests only: 1 = the candidate top-k of the detection tails (ivx_anchor_head_get_bboxes,
 * ivx_fcos_head_level_candidates) always runs as the one-workgroup radix select; 0 (default) = lists of >= 16 384 scores take
 * the chip-wide histogram / compaction form.  Both return the same indices in the same order. */
int ivx_topk_set_mode(int32_t single_workgroup);

/* ---------------------------------------------------------------------------------------
 * Device ceilings measured on the box (measurement only; bench.py prices its roofline fractions against the data-sheet
 * peaks AND these): the dense issue rate of the MFMA form the conv kernel uses for `dtype` (IVX_F32:
 * v_mfma_f32_32x32x2_f32, IVX_BF16: v_mfma_f32_32x32x16_bf16; scratch >= 512 KiB of device memory), and the streaming
 * copy rate of HBM (read + written bytes per second over `bytes` from src to dst; use buffers well past the 256 MiB
 * Infinity Cache).  Both synchronise the stream and return the best of a few repetitions. */
int ivx_ubench_mfma(int32_t dtype, void *scratch, int64_t scratch_bytes, double *tflops, ivx_stream_t stream);
int ivx_ubench_copy(const void *src, void *dst, int64_t bytes, double *gbps, ivx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IMVOXEL_LAB_H_ */
