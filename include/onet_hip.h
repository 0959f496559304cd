/*
 * onet_hip.h -- C ABI of libonet_hip.so: hand-written gfx950 (MI355X, CDNA4) HIP
 * kernels for the Onet twin-U-Net hot path.
 *
 * This is the drop-in boundary.  The reference (joeyee/Onet) is pure Python on
 * PyTorch; its "FFI" for this path is the set of ATen operators its model file
 * calls.  Every entry point below cites the reference call site it replaces
 * ("OV" = source_code/Onet_vanilla_20240606.py).  The Python mirror of the
 * reference's nn.Module surface (onet_amd/modules.py) binds these with ctypes.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    stated; no torch types, no C++ types, no exceptions across the boundary;
 *  - return 0 on success, a negative ONET_E* code otherwise; the message is
 *    available from onet_last_error() (thread-local);
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *    call only enqueues work -- no allocation, no host sync (graph-capturable);
 *  - tensors are fp32 NCHW.  Activations carry an explicit BATCH STRIDE in
 *    elements (`*_bs`): image b, channel c, pixel p lives at
 *        base + b*bs + c*H*W + p
 *    so a tensor may be a channel-slice of a wider concat buffer (OV:100 writes
 *    skip and upsampled halves side by side; we never copy for it);
 *  - "packed" weights are produced by onet_conv_pack_* once per optimizer step.
 */
#ifndef ONET_HIP_H
#define ONET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ONET_OK 0
#define ONET_EINVAL (-1)   /* bad argument / unsupported shape */
#define ONET_EHIP (-2)     /* HIP runtime error (launch failure, ...) */

const char* onet_last_error(void);
int onet_abi_version(void);            /* bumps when a signature changes */
int onet_device_info(int* cu_count, int* lds_bytes, int* wave_size, char* arch, int arch_len);

/* ---- weight packing --------------------------------------------------- */
/* nn.Conv2d weight [Cout][Cin][3][3] (OV:47,51) ->
 *   wp_fwd   [Cin ][9][Cout]            (fwd  implicit-GEMM A operand)
 *   wp_dgrad [Cout][9][Cin ] taps flipped (dgrad = conv of dZ with W^T rot180)
 * either output may be NULL. */
int onet_conv3x3_pack_weights(const float* w, float* wp_fwd, float* wp_dgrad,
                              int Cout, int Cin, void* stream);
/* nn.ConvTranspose2d weight [Cin][Cout][2][2] (OV:86) ->
 *   wp_fwd   [Cin][4*Cout]   column m = (dy*2+dx)*Cout + co   (1x1 conv to 4*Cout "sub-pixel" channels)
 *   wp_dgrad [4*Cout][Cin]   row    k = (dy*2+dx)*Cout + co */
int onet_convT2x2_pack_weights(const float* w, float* wp_fwd, float* wp_dgrad,
                               int Cin, int Cout, void* stream);
/* Fused forward of nn.ConvTranspose2d(Cin, Ct, kernel_size=2, stride=2) (OV:86) + F.pad offsets + the write into
 * the second half of the concat buffer (OV:91-100): one 1x1 MFMA GEMM whose epilogue does the pixel shuffle and
 * the bias.  wq [Cin][4*Ct] with column c*4 + (2*di + dj); y is the [Ct][Ho][Wo] window (batch stride y_bs) and
 * receives rows pt .. pt+2h-1, columns pl .. pl+2w-1 (the caller zeroes a non-empty F.pad border). */
int onet_convT2x2_pack_weights_fused(const float* w, float* wq, int Cin, int Cout, void* stream);
/* operand_bf16 (the forward, both backward GEMMs and their _b / _dbias forms): operand precision of THIS call on the 128 x 128
 * fast path -- 1 rounds the MFMA operands to bf16 (nearest-even) with fp32 accumulation and fp32 results, i.e.
 * nn.ConvTranspose2d under torch.autocast(bfloat16), BASELINE configs[2]; 0 = fp32 MFMA; 2 = fp32-level results on the bf16 matrix
 * cores by operand splitting (each operand = hi + mid bf16 parts, three MFMAs per term, as onet_conv3x3_split_*: the default of
 * the fp32 model since round 3).  Shapes outside the fast path run the fp32 kernels whatever the value.  (ABI 2: a per-call
 * argument; ABI 1 had a process-wide switch, which two models of different precision in one process could not share.) */
int onet_convT2x2_fwd(const float* x, int64_t x_bs, const float* wq, const float* bias, float* y, int64_t y_bs,
                      int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16,
        void* stream);
/* Backward of the same module without materialising the space-to-depth tensor: the 1x1 GEMM kernels gather their
 * dy operand straight from the [Ct][Ho][Wo] window of the concat gradient (dy_bs = its batch stride).
 *   dgrad: dx[b][ci][i][j] = sum_{c,di,dj} dy[b][c][pt+2i+di][pl+2j+dj] * W[ci][c][di][dj]   (wp_dgrad of
 *          onet_convT2x2_pack_weights; Ct % 8 == 0)
 *   wgrad: dw[ci][c][di][dj] = sum_{b,i,j} x[b][ci][i][j] * dy[b][c][pt+2i+di][pl+2j+dj]     (Ct % 64 == 0; workspace
 *          of onet_convT2x2_wgrad_ws_bytes(B, Cin, Ct, h, w))
 * Maps whose pixel count is a multiple of 128 (32 for wgrad) with Cin % 128 == 0, Ct % 32 == 0 and no F.pad offsets -- every
 * Up block of a 2^k-sized input -- run as DMA-fed 128 x 128 MFMA GEMMs (convt_gemm.hip), anything else on the 64-row
 * direct kernels.
 *   dbias: db[c] = sum dy[:, c, window]  (scratch: B*C doubles) */
int onet_convT2x2_dgrad(const float* dy, int64_t dy_bs, const float* wp_dgrad, float* dx, int64_t dx_bs, int B,
                        int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16,
        void* stream);
/* dx and dbias in one launch where the 128 x 128 GEMM path takes the shape (returns 1 and does nothing otherwise: use
 * onet_convT2x2_dgrad + onet_convT2x2_dbias); ws: onet_convT2x2_dgrad_dbias_ws_bytes() bytes of scratch */
int64_t onet_convT2x2_dgrad_dbias_ws_bytes(int B, int Ct, int h, int w);
int onet_convT2x2_dgrad_dbias(const float* dy, int64_t dy_bs, const float* wp_dgrad, float* dx, int64_t dx_bs, float* dbias,
                              void* ws, int64_t ws_bytes, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16,
        void* stream);
int64_t onet_convT2x2_wgrad_ws_bytes(int B, int Cin, int Ct, int h, int w);   /* workspace of onet_convT2x2_wgrad */
int onet_convT2x2_wgrad(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, float* dw, void* ws,
                        int64_t ws_bytes, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16,
        void* stream);
int onet_convT2x2_dbias(const float* dy, int64_t dy_bs, float* dbias, double* scratch, int accumulate, int B,
                        int C, int h, int w, int Ho, int Wo, int pt, int pl, void* stream);

/* ---- K1: 3x3 / 1x1 convolution, fp32 MFMA implicit GEMM ----------------- */
/* z[b][co][y][x] = sum_{ci,ky,kx} wp[ci][ky*ks+kx][co] * x[b][ci][y+ky-p][x+kx-p]
 * ks in {1,3}, p = ks/2, zero padding, stride 1, no bias (OV:47,51 via F.conv2d).
 * dgrad is the same call with wp_dgrad and Cin/Cout swapped.
 * bn_part (nullable): per-block partial sums for BatchNorm statistics,
 *   layout [nparts][Cout][3] floats (n, mean, M2), nparts from
 *   onet_conv_fwd_nparts(); consumed by onet_bn_finalize. */
int onet_conv_fwd(const float* x, int64_t x_bs, const float* wp, float* z, int64_t z_bs,
                  float* bn_part, int B, int Cin, int Cout, int H, int W, int ks, void* stream);
int onet_conv_fwd_nparts(int B, int Cout, int H, int W);

/* Fast fp32 path for ks=3, fwd / dgrad: Winograd F(4x4,3x3) on the fp32 matrix cores (4x fewer multiplies; 6x6 input tiles, interpolation
 * points 0, +-1, +-2; fp32 error ~3e-6 rms per layer).  wq_fwd [Cin][36][Cout], wq_dgrad [Cout][36][Cin].
 * Requires Cin % 4 == 0 and Cout % 4 == 0.  Same call sites as onet_conv_fwd (OV:47,51).  (Round 5: the F(2x2,3x3) kernels
 * of round 1 are gone; layers too small for these blocks take onet_conv_fwd / onet_conv_wgrad.) */
int onet_conv3x3_pack_weights_winograd4(const float* w, float* wq_fwd, float* wq_dgrad,
                                        int Cout, int Cin, void* stream);
int onet_conv3x3_winograd4_fwd(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs,
                               int B, int Cin, int Cout, int H, int W, void* stream);
/* The same convolution with the BatchNorm statistics pass of OV:48,52 folded into its epilogue: every block also
 * writes the (n, mean, M2) record of its 16 x 32 output pixels per channel, CHANNEL-MAJOR:
 * part [Cout][nparts][3], nparts = onet_conv3x3_winograd4_nparts(B, H, W) = B * blocks per image (image-major, so a
 * batch slice is a contiguous range of records); consumed by onet_bn_finalize_cm.  W > 16 only (nparts() returns 0
 * where the statistics are not emitted: use onet_bn_stats_partial there). */
int onet_conv3x3_winograd4_nparts(int B, int H, int W);
int onet_conv3x3_winograd4_fwd_stats(const float* x, int64_t x_bs, const float* wq, float* z, int64_t z_bs,
                                     float* part, int B, int Cin, int Cout, int H, int W, void* stream);
/* The F(4x4) input-gradient launch of a Conv-BN-ReLU-Conv chain (OV:47-53: da = dgrad(dz) is the gradient of the
 * ACTIVATION a = relu(bn(z_prev)) of the convolution below) with the first BatchNorm-backward pass folded into its
 * epilogue: part2 [Cda][nparts][2] = (sum dy, sum dy * xhat) per 16 x 32-pixel block, dy = da * [a > 0], xhat from
 * save_prev [B / group_images][4][Cda] (onet_bn_finalize's `save`, one per statistics group of group_images
 * consecutive images); nparts = onet_conv3x3_winograd4_nparts(B, H, W) (image-major), consumed by
 * onet_bn_bwd_finalize_cm instead of onet_bn_relu_bwd_reduce + onet_bn_bwd_finalize.  Full blocks only. */
int onet_conv3x3_winograd4_dgrad_bnreduce(const float* dz, int64_t dz_bs, const float* wq_dgrad, float* da,
                                          int64_t da_bs, const float* z_prev, int64_t z_prev_bs,
                                          const float* save_prev, int group_images, float* part2,
                                          int B, int Cdz, int Cda, int H, int W, void* stream);
/* nn.ConvTranspose2d(k=2,s=2) forward with the up-sampled tensor written pre-split (round 4: fp16 hi | mid slots [B][Ct/8][Ho][2][Wo][8], batch stride in 4-byte units), e.g.
 * into the up-sampled channel groups of a pre-split concat buffer; no fp32 output.  Returns 1 (nothing done) outside the GEMM fast path. */
int onet_convT2x2_fwd_p(const float* x, int64_t x_bs, const float* wq, const float* bias, void* yP, int64_t yP_bs, const void* y_amax,
                        int nparts, int B, int Cin, int Ct, int h, int w, int Ho, int Wo, int pt, int pl, int operand_bf16, void* stream);
/* Round 5 -- the same forward with BOTH operands in slot form: x pre-split by its producer ([B][Cin/8][h][nparts][w][8], batch stride in
 * 4-byte units; x_amax: the magnitude slots it was scaled by, guard rule, NULL = unscaled) and the weights packed once per optimizer step
 * by onet_convT2x2_pack_weights_slots into wP [Cin/8][nparts][4 Ct][8] (nparts = 2: fp16 hi | mid parts of 2^k w, (2^k, 2^-k) as two
 * floats behind the pack -- allocate Cin * 4 Ct * 2 * nparts + 8 bytes; amax_ws: 8 KB of scratch; nparts = 1: bf16(w); wdP (may be
 * NULL): the same launch also writes the input gradient's K-slot pack, below).  Every MFMA
 * fragment is one 16-byte LDS read of a DMA-copied slot: no conversion or split arithmetic in the kernel.  Output and y_amax as
 * onet_convT2x2_fwd_p.  Returns 1 (nothing done) unless Cin % 32 == 0, Ct % 32 == 0, h w % 128 == 0 and w is even. */
int onet_convT2x2_pack_weights_slots(const float* w, void* wP, void* wdP, void* amax_ws, int Cin, int Ct, int nparts, void* stream);
int onet_convT2x2_fwd_slots(const void* xP, int64_t xP_bs, const void* x_amax, const void* wP, const float* bias, void* yP, int64_t yP_bs,
                            const void* y_amax, int nparts, int B, int Cin, int Ct, int h, int w, void* stream);
/* ... and the backward GEMMs on slot operands.  dyP [B][Ct/8][2h][nparts][2w][8]: the up-sampled half of the concat gradient, written
 * pre-split by onet_conv3x3_split_dgrad_pre_slots as parts of 2^k dy (k by the `always` rule from the bound in dy_amax,
 * onet_conv3x3_dgrad_bound).  Input gradient: wdP = onet_convT2x2_pack_weights_slots' second pack [(Ct/8) 4][nparts][Cin][8] (K-slot =
 * 8 channels at one sub-pixel; same size and scale pair as the forward pack), dx fp32 [B][Cin][h][w].  Weight gradient: x pre-split as
 * for the forward; dw [Cin][Ct][2][2]; dbias (may be NULL) [Ct] = sum of dy over all pixels, taken in the same launch; ws: at least
 * onet_convT2x2_wgrad_slots_ws_bytes(...) (0: shape not taken).  Both return 1 (nothing done) outside Cin % 128 == 0, Ct % 32 == 0,
 * h w % 128 == 0 (weight gradient: w a power of two). */
int onet_convT2x2_dgrad_slots(const void* dyP, int64_t dyP_bs, const void* dy_amax, const void* wdP, float* dx, int64_t dx_bs, int nparts, int B,
                              int Cin, int Ct, int h, int w, void* stream);
int64_t onet_convT2x2_wgrad_slots_ws_bytes(int B, int Cin, int Ct, int h, int w);
int onet_convT2x2_wgrad_slots(const void* xP, int64_t xP_bs, const void* x_amax, const void* dyP, int64_t dyP_bs, const void* dy_amax, float* dw,
                              float* dbias, void* ws, int64_t ws_bytes, int nparts, int B, int Cin, int Ct, int h, int w, void* stream);
/* y_amax (may be NULL: unscaled): magnitude slots holding a bound of the up-sampled tensor, written by onet_convT2x2_out_bound from the
 * weights (nn.ConvTranspose2d layout [Cin][Ct][2][2]), the bias and the exact max |x| (x_amax, recorded by the pass that wrote x): the
 * fp16 parts are those of 2^k y with the guard exponent the slots select. */
int onet_convT2x2_out_bound(const float* w, const float* bias, int Cin, int Ct, const void* x_amax, void* y_amax, void* stream);

/* ---- fp32 3x3 convolution on the bf16 matrix cores by OPERAND SPLITTING (conv_split.hip; F.conv2d + input gradient, OV:47,51).
 * x = x_hi + x_mid, w = w_hi + w_mid with bf16 parts (hi = round-to-nearest-even, mid = bf16 of the remainder: 16 significant
 * bits together); z = conv(x_hi, w_hi) + conv(x_hi, w_mid) + conv(x_mid, w_hi) on v_mfma_f32_32x32x16_bf16 with fp32
 * accumulation: fp32 tensors in, fp32 out, error of the split 1e-6 rms of the output scale (below the fp32 Winograd F(4x4)
 * kernel's), three MFMAs per product term at 16x the fp32 MFMA rate.  wq_fwd [Cin/16][2 parts][9][2][Cout][8] bf16 (2 * 9 * Cin *
 * Cout elements), wq_dgrad [ceil(Cout/16)][2][9][2][Cin][8] with the taps rotated; either may be NULL.  Requires Cin % 16 == 0 and
 * W > 16.  _fwd_stats also writes one BatchNorm record (n, mean, M2) per channel and 16 x 32-pixel tile: part [Cout][nparts][3],
 * nparts = onet_conv3x3_split_nparts(B, H, W) (0: the map is not made of full tiles).
 * fp16 parts (fwd_f16 / wq_f16 = 1; the forward convolution's default): hi = fp16(v), mid = fp16(v - hi) carry 22 significant bits
 * instead of 16 at the same three MFMAs per term (v_mfma_f32_32x32x16_f16): error at the fp32 direct kernel's level.  The pack
 * holds 2^8 w (undone on the accumulators); the input must stay within fp16's range (|x| < 65504: activations, not gradients --
 * an input gradient is computed with wq_dgrad, whose parts are always bf16, and wq_f16 = 0). */
int onet_conv3x3_split_pack_weights(const float* w, void* wq_fwd, void* wq_dgrad, void* amax_ws, int Cout, int Cin, int fwd_f16,
                                    int dgrad_f16, void* stream);
/* ABI 3 (round 4) -- fp16 parts with per-tensor power-of-two scales.  An fp16 pack holds the parts of 2^k w with k chosen from the
 * tensor's largest magnitude (amax_ws: 8 KB of device scratch) and carries (2^k, 2^-k) as two floats BEHIND the pack (buffers:
 * 2 * K * 9 * N 16-bit elements + 16 bytes).  Activations / gradients carry their largest magnitude in 64 "magnitude slots"
 * (unsigned[64 * 32]: one slot per 128-byte line, fp32 bit patterns, maximum over the slots; written with atomicMax by the producers: onet_bn_relu_apply,
 * onet_bn_relu_bwd_apply).  _conv_amax: forward (scale_always = 0: x is scaled only when its magnitude would leave fp16's
 * range -- a guard) or input gradient (scale_always = 1, the dgrad pack: the operand's amax is brought to [2^13, 2^14)) of a 3x3
 * convolution on fp16 parts, three v_mfma_f32_32x32x16_f16 per term; x_amax = NULL: unscaled.  part as _fwd_stats.
 * _wgrad_f16: the weight gradient on fp16 parts of x (guard, may be NULL) and dz (always scaled; required); save != NULL:
 * normalise on load as _wgrad_norm. */
/* max |x| over n contiguous floats into 64 magnitude slots (atomicMax: the slots must be zeroed, or hold an earlier maximum) */
int onet_absmax_slots(const float* x, int64_t n, void* slots, void* stream);
int onet_conv3x3_split_conv_amax(const float* x, int64_t x_bs, const void* x_amax, int scale_always, const void* wq, float* z, int64_t z_bs,
                                 float* part, int B, int Cin, int Cout, int H, int W, void* stream);
int onet_conv3x3_split_wgrad_f16(const float* x, int64_t x_bs, const void* x_amax, const float* save, int n_groups, const float* dz,
                                 int64_t dz_bs, const void* dz_amax, float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout,
                                 int H, int W, int accumulate, void* stream);
int onet_conv3x3_split_fwd(const float* x, int64_t x_bs, const void* wq, int wq_f16, float* z, int64_t z_bs, int B, int Cin, int Cout,
                           int H, int W, void* stream);
int onet_conv3x3_split_nparts(int B, int H, int W);
/* Round 4 -- PRE-SPLIT operands.  The activation of a 3x3 convolution (OV:47,51) arrives already split, in the slot layout the
 * MFMA tile wants:  xs [B][C/8][H][part 2][W][8] 16-bit (part 0 = hi, 1 = mid; fp16 when f16 != 0, else bf16): 4 bytes per element,
 * the fp32 tensor's footprint; batch stride xs_bs in 4-byte units.  The producers write it (BatchNorm + ReLU apply / pooling, the
 * BatchNorm backward apply, the ConvTranspose2d epilogue); onet_split_pack_act converts an fp32 NCHW tensor (times `scale`, a power
 * of two) for tests and for producers without a fused variant.  _fwd_pre: z = conv(xs, wq) with the weight pack of
 * onet_conv3x3_split_pack_weights (same arithmetic as onet_conv3x3_split_fwd: bit-identical results for the same parts); staging
 * is an LDS-DMA copy.  x_amax / scale_always: the magnitude slots and rule (amax_scale) the PRODUCER scaled xs by -- the kernel
 * undoes that power of two on its accumulators; NULL: xs is unscaled.  x_amax2 / split_ch: the input is a concat buffer whose
 * channels >= split_ch (0: none) were scaled by a second producer with its own slots.  part != NULL: BatchNorm statistics records
 * as onet_conv3x3_split_fwd_stats. */
int onet_split_pack_act(const float* x, int64_t x_bs, void* xs, int64_t xs_bs, int B, int C, int H, int W, int f16, float scale,
                        const void* slots /* NULL | magnitude slots: their guard scale (amax_scale, not always) multiplies `scale` */,
                        void* stream);
/* Round 5 -- input gradient of the SECOND convolution of a DoubleConv (OV:51's backward) from pre-split dz, with the BatchNorm-backward
 * reduce pass of the FIRST unit (OV:48-49's backward: sum dy, sum dy xhat per channel, dy = da where the unit's output was positive)
 * taken from the accumulator tile in the epilogue -- this launch's output IS that unit's da.  z_prev / save: the first unit's
 * pre-activation [B][Cout][H][W] fp32 and coefficients [G][4][Cout] (group_images per statistics group; 0: one group); rec4
 * [onet_conv3x3_split_pre_nparts(B, H, W)][Cout][4]: records in onet_bn_relu_bwd_reduce's format, ready for onet_bn_bwd_finalize;
 * da_amax (may be NULL): magnitude slots that receive max |da|.  Here Cin = channels of dz (the reduction), Cout = channels of da.
 * Returns 1, nothing launched, where the shape is not taken (maps not made of full 16 x 32 tiles; 16-pixel maps with (hi | mid) parts;
 * fewer than four K chunks per tile, where the epilogue's z loads are not covered). */
int onet_conv3x3_split_dgrad_pre_bnreduce(const void* dzs, int64_t dzs_bs, const void* dz_amax, int scale_always, const void* wq, int wq_f16,
                                          float* da, int64_t da_bs, const void* z_prev, int z_bf16, int64_t z_bs, const float* save, int group_images,
                                          float* rec4, void* da_amax, int B, int Cin, int Cout, int H, int W, void* stream);
int onet_conv3x3_split_fwd_pre(const void* xs, int64_t xs_bs, const void* x_amax, int scale_always, const void* x_amax2, int split_ch,
                               const void* wq, int wq_f16, void* z, int z_bf16 /* 1: z stored as bf16 (plain-bf16 operands only) */, int64_t z_bs,
                               float* part, int B, int Cin, int Cout, int H, int W,
                               void* stream);
/* Round 5 -- the input gradient of a decoder block's FIRST convolution (fp16 hi | mid parts, maps made of full 16 x 32 tiles): channels
 * < ch0 of da (the skip half of the concat gradient) as fp32, channels >= ch0 (the up-sampled half, read only by the ConvTranspose2d
 * backward GEMMs) pre-split into daP [B][(Cout - ch0)/8][H][2][W][8] as parts of 2^k da, k by the `always` rule from daP_amax -- which
 * onet_conv3x3_dgrad_bound fills first: max |dz| (dz_amax) x the largest sum over (co, tap) of |w[co][ci][tap]| over ci >= ci0 (w: the
 * nn.Conv2d weight [Cout][Cin][3][3]; out_slots zeroed by the caller).  Cin / Cout here name the channels of dzs / da.  wq_f16 = 2:
 * plain bf16 operands -- daP [B][(Cout - ch0)/8][H][W][8] is one part of bf16(da), unscaled (daP_amax may be NULL). */
int onet_conv3x3_split_dgrad_pre_slots(const void* dzs, int64_t dzs_bs, const void* dz_amax, int scale_always, const void* wq, int wq_f16,
                                       float* da, int64_t da_bs, void* daP, int64_t daP_bs, int ch0, const void* daP_amax, int B, int Cin,
                                       int Cout, int H, int W, void* stream);
int onet_conv3x3_dgrad_bound(const float* w, int Cout, int Cin, int ci0, const void* dz_amax, void* out_slots, void* stream);
/* Weight gradient (OV:47,51 backward) from pre-split x and dz (both in the slot layout, same 16-bit type): fragments by the gfx950
 * transposing LDS read, staging by LDS-DMA; the producers' power-of-two scales (x_amax: guard rule, dz_amax: always; NULL:
 * unscaled) are undone on the slabs; deterministic split-K through ws
 * (onet_conv3x3_split_wgrad_ws_bytes).  _ok: W >= 64, or W = 32 / 16 with the batch a multiple of 2 / 4; Cin, Cout multiples of 8.
 * _fwd_pre also takes maps exactly 16 pixels wide (even batch, H % 16 == 0): two images side by side per tile. */
int onet_conv3x3_split_pre_nparts(int B, int H, int W);    /* statistics records of _fwd_pre (16-pixel-wide maps: one per image pair) */
int onet_conv3x3_split_wgrad_pre_ok(int B, int Cin, int Cout, int H, int W);
int onet_conv3x3_split_wgrad_pre(const void* xs, int64_t xs_bs, const void* x_amax, const void* x_amax2, int split_ch, const void* dzs,
                                 int64_t dzs_bs, const void* dz_amax, int f16, float* dw, void* ws, int64_t ws_bytes, int B, int Cin,
                                 int Cout, int H, int W, int accumulate, void* stream);
int onet_conv3x3_split_fwd_stats(const float* x, int64_t x_bs, const void* wq, int wq_f16, float* z, int64_t z_bs, float* part, int B,
                                 int Cin, int Cout, int H, int W, void* stream);
/* Weight gradient of the same convolution with both operands (x, dz: fp32 NCHW) split the same way, three MFMAs per term;
 * row-streaming units (one image row of a 64-pixel strip), deterministic split-K through ws (onet_conv3x3_split_wgrad_ws_bytes).
 * _ok: 1 where the kernel takes the shape: W >= 64 with W % 4 == 0, or W = 32 / 16 with B a multiple of 64 / W (that many images
 * side by side make one unit; the whole batch of x and of dz must then lie within 2 GiB); 16-byte aligned image rows (x_bs,
 * dz_bs % 4 == 0). */
int onet_conv3x3_split_wgrad_ok(int B, int Cin, int Cout, int H, int W);
/* Normalise on load (the second convolution of a DoubleConv, OV:51, without a materialised activation of the first, OV:49):
 * z_prev = the PRE-activation of the Conv-BatchNorm-ReLU unit below, save = its BatchNorm coefficients [n_groups][4][Cin] (rows
 * mean, invstd, scale, shift as onet_bn_train_coeffs writes them; statistics groups = n_groups equal runs of consecutive
 * images).  The staging threads compute max(fma(z - mean, scale, shift), 0) -- bit for bit what onet_bn_relu_apply writes --
 * before splitting; zero padding stays zero.  _fwd_norm: part != NULL also emits the statistics records (as _fwd_stats);
 * _wgrad_norm: n_groups 1 or 2.  Results are bit-identical to the plain entry points on the materialised activation. */
int onet_conv3x3_split_fwd_norm(const float* z_prev, int64_t z_bs, const float* save, int n_groups, const void* wq, int wq_f16,
                                float* z, int64_t zo_bs, float* part, int B, int Cin, int Cout, int H, int W, void* stream);
int onet_conv3x3_split_wgrad_norm(const float* z_prev, int64_t z_bs, const float* save, int n_groups, const float* dz, int64_t dz_bs,
                                  float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int accumulate,
                                  void* stream);
int64_t onet_conv3x3_split_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W);
int onet_conv3x3_split_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws, int64_t ws_bytes,
                             int B, int Cin, int Cout, int H, int W, int accumulate, void* stream);

/* The stem: first 3x3 convolution of the U-Net (F.conv2d of `inc`, OV:47, Cin = n_channels in 1..4, no bias) and the batch
 * statistics of its BatchNorm (OV:48) in one streaming pass.  w: the nn.Conv2d weight [Cout][Cin][3][3] as it is (no pack);
 * part [Cout][nparts][3] = (n, mean, M2) per 16 x 64-pixel tile for onet_bn_finalize_cm, or NULL (convolution only).
 * nparts = onet_conv3x3_stem_nparts(B, Cin, Cout, H, W); 0 = not applicable (Cin > 4, Cout > 128, H % 16 or W % 64): use
 * onet_conv_fwd + onet_bn_stats_partial.
 * ABI 4 -- K7 fused (OV:180): twin_B > 0 says the batch of B = 2 twin_B images is the twin batch
 * [X ; clip(1 - X + twin_bias, 0, 1)] of which only X exists (x holds twin_B images): image b >= twin_B is formed from image
 * b - twin_B while the halo tile is loaded, with onet_complement_clip's expression.  twin_B = 0: an ordinary batch of B images. */
int onet_conv3x3_stem_nparts(int B, int Cin, int Cout, int H, int W);
int onet_conv3x3_stem_fwd_stats(const float* x, int64_t x_bs, const float* w, float* z, int64_t z_bs, float* part, int B, int Cin,
                                int Cout, int H, int W, int twin_B, float twin_bias, void* stream);

/* Larger-tile weight gradient: Winograd F(3x3,4x4) (4x fewer multiplies than direct, 1.78x fewer than the F(2x2,3x3)
 * kernel below; 6x6 input tiles, 4x4 tiles of dz in the filter's role, the points of onet_conv3x3_winograd4_fwd; split-K
 * 36-position slabs folded by A'^T . A').  Where onet_conv3x3_winograd4_wgrad_ok() returns 1 (W % 32 == 0, H % 4 == 0,
 * Cin % 32 == 0); dW = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G, split-K, deterministic slab reduction; dw is [Cout][Cin][3][3]. */
int onet_conv3x3_winograd4_wgrad_ok(int B, int Cin, int Cout, int H, int W);
int64_t onet_conv3x3_winograd4_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W);
int onet_conv3x3_winograd4_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw, void* ws,
                                 int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int accumulate, void* stream);

/* wgrad: dw[co][ci][ky][kx] (+)= sum_{b,y,x} dz[b][co][y][x] * x[b][ci][y+ky-p][x+kx-p]
 * (autograd of F.conv2d wrt weight; reference: implicit via loss.backward(), TS:217).
 * Two launches: split-K partial slabs into `ws`, then a deterministic slab
 * reduction.  ws must hold onet_conv_wgrad_ws_bytes() bytes.
 * out_layout 0: dw is [Cout][Cin][ks][ks] (nn.Conv2d);
 * out_layout 1: dw is nn.ConvTranspose2d [Cx][Cdz/4][2][2] where the dz operand has
 *               4*Ct sub-pixel channels (q*Ct+co) and x operand is the convT input. */
int onet_conv_wgrad(const float* x, int64_t x_bs, const float* dz, int64_t dz_bs, float* dw,
                    void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H, int W, int ks,
                    int out_layout, int accumulate, void* stream);
/* Round 4: the stem's weight gradient (Cin <= 4: nn.Conv2d(n_channels, 64, 3) of OV:111, whose input has no gradient) with the
 * BatchNorm + ReLU backward of its unit applied on load: dz is not materialised -- the kernel computes it per element from da (the
 * gradient of the unit's output), the pre-activation z and the layer's coefficients (save, coef [G][4][Cout]; group_images as in the
 * BatchNorm entry points) with onet_bn_relu_bwd_apply's arithmetic: the same bits as that pass followed by onet_conv_wgrad.
 * Workspace: onet_conv_wgrad_ws_bytes(..., ks = 3).  twin_B / twin_bias: as onet_conv3x3_stem_fwd_stats (x = the first half of the
 * twin batch; the complement half is formed on load). */
int onet_conv3x3_stem_wgrad_bn(const float* x, int64_t x_bs, const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save,
                               const float* coef, int group_images, float* dw, void* ws, int64_t ws_bytes, int B, int Cin, int Cout, int H,
                               int W, int accumulate, int twin_B, float twin_bias, void* stream);
int64_t onet_conv_wgrad_ws_bytes(int B, int Cin, int Cout, int H, int W, int ks);

/* ---- K2/K3: BatchNorm2d (+ReLU) ---------------------------------------- */
/* Welford partials of z per channel: part [nparts][C][3] = (n_k, mean_k, M2_k = sum (z-mean_k)^2)
 * over nparts = B*chunks slices of the B*HW values (standalone statistics pass; the conv epilogue
 * can emit the same records instead).  Robust where E[z^2]-mean^2 is not (tiny counts, |mean|>>std). */
int onet_bn_stats_partial(const float* z, int64_t z_bs, float* part, int nparts,
                          int B, int C, int HW, void* stream);
/* train-mode finalize (OV:48,52 -> F.batch_norm(training=True)):
 *   Chan merge of the partials in fp64: mean, var = M2/N (biased), invstd = rsqrt(var+eps)
 *   scale = gamma*invstd
 *   running_mean = (1-m)*running_mean + m*mean
 *   running_var  = (1-m)*running_var  + m*var*N/(N-1)     (unbiased, verified SURVEY §8c)
 * save = [4][C]: mean, invstd, scale, beta.  running_* may be NULL.
 * act_amax (may be NULL): 64 magnitude slots that receive the bound |gamma| sqrt(N - 1) + |beta| of the activation relu(bn(z)) -- what
 * the pre-split producers below scale by (round 5: the former _act forms are these entry points). */
int onet_bn_finalize(const float* part, int nparts, int64_t count, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, float momentum, float eps, float* save, void* act_amax, int C, void* stream);
/* onet_bn_finalize for channel-major partials: channel c's records are part[c * c_stride + 3 * p], p < groups * nparts (groups >= 1
 * statistics groups, see below; c_stride >= groups * nparts * 3). */
int onet_bn_finalize_cm(const float* part, int nparts, int64_t c_stride, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, float* save, void* act_amax, int groups, int C, void* stream);
/* eval-mode coefficients from running stats: same `save` layout. */
int onet_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* save, int C, void* stream);
/* a = max(0, (z-mean)*scale + beta)   (BN + nn.ReLU, OV:48-49).  z_bf16: z is stored as bf16 (ABI 4, below).  amax (may be NULL): 64
 * magnitude slots (zeroed by the caller; the statistics groups of a twin batch share them) that receive max a -- the overflow guard
 * of the fp16-split convolution that consumes the activation.  group_images: statistics groups, below. */
int onet_bn_relu_apply(const void* z, int z_bf16, int64_t z_bs, float* a, int64_t a_bs, const float* save, void* amax, int group_images,
                       int B, int C, int HW, void* stream);
/* relu(bn(z)) AND its 2x2 max-pooling in one pass (nn.BatchNorm2d + nn.ReLU of an encoder block followed by nn.MaxPool2d(2),
 * OV:48-49 / 52-53 -> OV:67): a [B,C,H,W], y [B,C,H/2,W/2].  Bit-identical to onet_bn_relu_apply followed by onet_maxpool2_fwd; amax
 * as above (the pooled tensor has the same maximum).  Returns 1 (nothing done) unless H % 2 == 0, W % 4 == 0 and rows are 16-byte
 * aligned. */
int onet_bn_relu_apply_pool(const float* z, int64_t z_bs, float* a, int64_t a_bs, float* y, int64_t y_bs, const float* save,
                            void* amax, int B, int C, int H, int W, void* stream);

/* backward of BN(train)+ReLU.  dy = da * ((z-mean)*scale+beta > 0);
 * pass 1: part2 [nparts][C][4] = (sum dy, sum dy*xhat) as (hi, lo) float pairs of fp64 sums (nparts a multiple of B: the records of
 *         image b's chunks are rows b * chunks .., so a statistics group's records are consecutive rows); da_amax (may be NULL): 64
 *         magnitude slots that receive max |da|;
 * finalize: dgamma (+)= sum dy*xhat, dbeta (+)= sum dy, coef [4][C] = (hi, lo) pairs of
 *           c1 = sum dy / N and c2 = sum dy*xhat / N; nparts, count: per group; part2 [groups][nparts][C][4], coef [groups][4][C];
 *           dz_amax (may be NULL, then save / da_amax may be NULL too): receives the bound of |dz| (as onet_bn_bwd_bound) -- da_amax
 *           must be complete, i.e. every reduce launch of the tensor precedes the first finalize;
 * pass 2: dz = scale * (dy - c1 - xhat*c2)     [train]   or  dz = scale*dy [eval: coef NULL],
 *         evaluated in fp64 per element like ATen's CPU kernel (accscalar_t = double); amax (may be NULL): 64 magnitude slots that
 *         receive max |dz|, what the fp16-split gradient kernels scale dz by (onet_conv3x3_split_conv_amax, onet_conv3x3_split_wgrad_f16). */
int onet_bn_relu_bwd_reduce(const float* da, int64_t da_bs, const void* z, int z_bf16, int64_t z_bs, const float* save, float* part2,
                            int nparts, void* da_amax, int group_images, int B, int C, int HW, void* stream);
int onet_bn_bwd_finalize(const float* part2, int nparts, int64_t count, float* dgamma, float* dbeta, float* coef, int accumulate,
                         int groups, int C, const float* save, const void* da_amax, void* dz_amax, void* stream);
/* onet_bn_bwd_finalize for the channel-major two-float records of onet_conv3x3_winograd4_dgrad_bnreduce:
 * channel c's records are part2[c * c_stride + 2 * p], p < nparts. */
int onet_bn_bwd_finalize_cm(const float* part2, int nparts, int64_t c_stride, int64_t count, float* dgamma,
                            float* dbeta, float* coef, int accumulate, int C, void* stream);
int onet_bn_relu_bwd_apply(const float* da, int64_t da_bs, const float* z, int64_t z_bs, const float* save, const float* coef,
                           float* dz, int64_t dz_bs, void* amax, int group_images, int B, int C, int HW, void* stream);
/* ---- Round 4, pre-split producers (bn.hip).  The BatchNorm + ReLU passes (OV:48-49, 52-53 and their backward) write their result
 * straight in the operand form of the split-fp16 convolution that consumes it: fp16 (hi, mid) parts in the slot layout
 * [B][C/8][H][2][W][8] (onet_split_pack_act's; batch strides in 4-byte units; C % 8 == 0).  Values are bit for bit those of the fp32
 * passes.  _apply_split: xs (+ the fp32 tensor a when not NULL).  _apply_pool_split: the activation to xs and / or a, the
 * 2 x 2-pooled values to ys (pre-split) or y (fp32); returns 1 when the shape is not taken (odd H / W, alignment).
 * _bwd_apply_split: dz as parts of 2^k dz, k chosen (conv_split.hip: amax_scale, always) from the magnitude slots dz_amax, which
 * must hold an upper bound of |dz| BEFORE the launch: onet_bn_bwd_bound writes it from the layer's coefficients and the exact
 * max |da| (da_amax: recorded by onet_bn_relu_bwd_reduce or onet_absmax_slots); the consumers read the same slots.
 * nparts = 2: the fp16 (hi | mid) slots above; nparts = 1: PLAIN bf16 operands, one part -- [B][C/8][H][W][8] bf16, rounded to
 * nearest even, unscaled (dz_amax may be NULL) -- for BASELINE configs[2]'s bf16 MFMA conv path (the same LDS-DMA staged kernels
 * with one part: wq_f16 / f16 = 2 in onet_conv3x3_split_fwd_pre / _wgrad_pre / _pack_weights / onet_split_pack_act).
 * STATISTICS GROUPS IN ONE LAUNCH (group_images / groups in the entry points below): the weight-shared twin batch [X ; 1 - X] is one
 * tensor whose halves are normalised separately (OV:178-179 runs the U-Net twice).  group_images > 0: images g * group_images ..
 * (g + 1) * group_images - 1 of the batch form group g and save / coef are [G][4][C]; 0: one group, save / coef [4][C].  The
 * finalize entry points take `groups`: records of group g follow those of group g - 1 (nparts records each), running statistics and
 * dgamma / dbeta take the groups' contributions in order -- the same bits as one launch per group.
 * ABI 4 -- z_bf16 (the entry points that READ the convolution output z): 0 = z is fp32; 1 = z is stored as bf16 [B][C][H][W] (BASELINE
 * configs[2], conv == "bf16": onet_conv3x3_split_fwd_pre rounds z once in its epilogue, the statistics come from its fp32
 * accumulators) -- batch strides stay in ELEMENTS, rows 8-byte aligned; every pass that reads z then moves half the bytes. */
int onet_bn_relu_apply_split(const void* z, int z_bf16, int64_t z_bs, void* xs, int64_t xs_bs, float* a, int64_t a_bs, const float* save,
                             const void* act_amax, int nparts, int group_images, int B, int C, int H, int W, void* stream);
int onet_bn_relu_apply_pool_split(const void* z, int z_bf16, int64_t z_bs, void* xs, int64_t xs_bs, float* a, int64_t a_bs, void* ys, int64_t ys_bs,
                                  float* y, int64_t y_bs, const float* save, const void* act_amax, int nparts, int group_images, int B, int C,
                                  int H, int W, void* stream);
/* act_amax (may be NULL: unscaled): the activation's magnitude slots holding the bound |gamma| sqrt(N - 1) + |beta| written by
 * onet_bn_finalize / _cm; the fp16 parts are those of 2^k a with the GUARD
 * exponent the slots select (0 unless the bound reaches 2^15: a loaded checkpoint with a huge gamma), undone by the consumers, which
 * read the same slots.  A concat buffer has one set of slots per producer (skip groups / up-sampled groups). */
int onet_bn_bwd_bound(const float* save, const float* coef, const void* da_amax, int64_t count, void* dz_amax, int C, void* stream);
int onet_bn_relu_bwd_apply_split(const float* da, int64_t da_bs, const void* z, int z_bf16, int64_t z_bs, const float* save, const float* coef,
                                 void* dzs, int64_t dzs_bs, const void* dz_amax, int nparts, int group_images, int B, int C, int H, int W,
                                 void* stream);

/* ---- K4: MaxPool2d(2) (OV:67) ------------------------------------------- */
int onet_maxpool2_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs,
                      int B, int C, int H, int W, void* stream);
/* dx = route dy to the first maximum of each 2x2 window (recomputed from x); rows/cols
 * beyond 2*floor(H/2) get 0.  accumulate: dx += instead of dx = (add must be NULL then).
 * add [+ add2] (nullable): dx = maxpool2 backward + add [+ add2] -- the gradient of a tensor that feeds BOTH the pooling and a skip
 * connection (OV:141-149: x1..x4 go to the next Down and to an Up's torch.cat; x1 is also returned, OV:152, and enters the head:
 * add2) in one pass instead of autograd's separate full-tensor adds. */
int onet_maxpool2_bwd(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                      const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, int B, int C, int H, int W, int accumulate,
                      void* stream);
/* onet_maxpool2_bwd (with add / add2) for an x that is the OUTPUT a = relu(bn(z)) of a Conv-BN-ReLU unit (OV:58 -> 67): the sum it
 * writes is that unit's activation gradient, so the unit's first BatchNorm-backward pass rides along: part2
 * [B * bands][C][4] = onet_bn_relu_bwd_reduce's records (bands = onet_maxpool2_bwd_bn_bands(H, W), 0 = fused form not
 * available), save [B / group_images][4][C].  add / add2 nullable. */
int onet_maxpool2_bwd_bn_bands(int H, int W);
/* x may be NULL (pre-split storage: x = relu(bn(z)) is recomputed from z and the coefficients, the same bits); dx_amax (may be NULL): 64
 * magnitude slots that receive max |dx| (dx is the producing layer's activation gradient: onet_bn_bwd_bound); z_bf16: ABI 4, above. */
int onet_maxpool2_bwd_add_bnreduce(const float* x, int64_t x_bs, const float* dy, int64_t dy_bs, const float* add, int64_t add_bs,
                                        const float* add2, int64_t add2_bs, float* dx, int64_t dx_bs, const void* z, int z_bf16, int64_t z_bs,
                                        const float* save, int group_images, float* part2, void* dx_amax, int B, int C, int H, int W,
                                        void* stream);

/* ---- K5/K6: ConvTranspose2d(k=2,s=2) pixel shuffle + pad + concat (OV:86-100) ---- */
/* sub [B][4*C][h][w] (1x1-conv output, channel q*C+co, q=dy*2+dx) + bias ->
 * y[b][co][pt+2y+dy][pl+2x+dx]; the rest of the Ho x Wo plane is zero-filled (F.pad). */
int onet_pixel_shuffle2_bias(const float* sub, const float* bias, float* y, int64_t y_bs,
                             int B, int C, int h, int w, int Ho, int Wo, int pt, int pl, void* stream);
/* inverse gather for backward: sub[b][q*C+co][y][x] = dy[b][co][pt+2y+dy][pl+2x+dx];
 * dy planes are Ho x Wo; dbias (nullable) (+)= sum over b,y,x,q (needs scratch: B*C doubles). */
int onet_space_to_depth2(const float* dy, int64_t dy_bs, float* sub, float* dbias, double* scratch,
                         int accumulate, int B, int C, int h, int w, int Ho, int Wo, int pt, int pl,
                         void* stream);
/* strided batch copy of a [B][n] block (torch.cat first half, OV:100) */
int onet_copy_strided(const float* src, int64_t src_bs, float* dst, int64_t dst_bs,
                      int B, int64_t n, void* stream);

/* ---- K5': bilinear x2, align_corners=True (OV:83) ------------------------- */
int onet_bilinear2x_fwd(const float* x, int64_t x_bs, float* y, int64_t y_bs, int B, int C,
                        int h, int w, int Ho, int Wo, int pt, int pl, void* stream);
/* gather form (ABI 4): every dx element sums the <= 6 x 6 output pixels whose stencil touches it, in a fixed order -- no atomics,
 * bitwise reproducible, dx needs no initialisation */
int onet_bilinear2x_bwd(const float* dy, int64_t dy_bs, float* dx, int64_t dx_bs, int B, int C,
                        int h, int w, int Ho, int Wo, int pt, int pl, void* stream);

/* ---- K7: Xd = clip(1 - X + bias, 0, 1) (OV:180) ---------------------------- */
int onet_complement_clip(const float* x, float* y, float bias, int64_t n, void* stream);

/* ---- K8/K9: head einsum + 2-way softmax (OV:176-189) ----------------------- */
/* Vt = sum_c Lt*Ht, Vd = sum_c Ld*Hd, S = softmax([Vt,Vd]) -> S [B][2][HW].
 * sLt, sLd (both or neither; [B][HW]): the per-pixel channel sums sL = sum_c L[c] as a by-product -- the only thing
 * jensen_shannon_divergence (OV:221-235) needs of L.
 * h_save_t, h_save_d (both or neither, round 5): Ht / Hd then hold the PRE-ACTIVATION z of the network's last Conv-BatchNorm-ReLU unit
 * and these are its coefficients [4][C] (onet_bn_finalize's `save`) for the two statistics groups: H = relu(bn(z)) is formed on load,
 * bit for bit what onet_bn_relu_apply writes -- the last activation tensor (OV:53 of up4) is never materialised. */
int onet_head_softmax_fwd(const float* Lt, int64_t Lt_bs, const float* Ht, int64_t Ht_bs, const float* Ld, int64_t Ld_bs, const float* Hd,
                          int64_t Hd_bs, float* Vt, float* Vd, float* S, float* sLt, float* sLd, const float* h_save_t, const float* h_save_d,
                          int B, int C, int HW, void* stream);
/* given dVt, dVd, dS (each nullable) and S: total dV, then dL = dV*H, dH = dV*L per branch; gsLt / gsLd (NULL = 0): the gradients of
 * the channel sums, folded into dLt / dLd, so that the loss neither re-reads L nor costs a full-tensor gradient add; h_save_*: as above
 * (dH is then the gradient of the unit's OUTPUT, which its BatchNorm backward takes as usual). */
int onet_head_softmax_bwd(const float* dVt, const float* dVd, const float* dS, const float* gsLt, const float* gsLd, const float* S,
                          const float* Lt, int64_t Lt_bs, const float* Ht, int64_t Ht_bs, const float* Ld, int64_t Ld_bs, const float* Hd,
                          int64_t Hd_bs, float* dLt, float* dHt, float* dLd, float* dHd, const float* h_save_t, const float* h_save_d, int B,
                          int C, int HW, void* stream);

/* ---- K10: JSD loss with the reference's log1pexp quirk (OV:221-267) ---------- */
/* One jsd term, Onet.jensen_shannon_divergence(Li, Si, Sprime) (OV:221-235):
 *   jsd = -mean_{b,p} f(-Si*sL) - mean_{b,p} f(Sp*sL),   sL = sum_c L[c]   (the p=64 x p=1 einsum
 *   broadcast of OV:231-232), f = the effective log1pexp of OV:237-251 (x <= -37 -> ln 2).
 * Si/Sp: [B][HW] with batch stride (slices of S).  sums [B*HW] saves sL for backward;
 * part: >= onet_jsd_nparts() doubles of scratch; jsd: 1 float.
 * compute_loss (OV:253-267) = -(jsd(Lt,St,Sd) + jsd(Ld,Sd,St)) / 2 is composed by the caller. */
/* (L may be NULL: `sums` is then an INPUT, the channel sums from onet_head_softmax_fwd) */
int onet_jsd_fwd(const float* L, int64_t L_bs, const float* Si, int64_t Si_bs,
                 const float* Sp, int64_t Sp_bs, float* sums, double* part, float* jsd,
                 int B, int C, int HW, void* stream);
int onet_jsd_nparts(void);
/* gL [B][HW]: d jsd / dL (identical for every channel -- caller broadcasts); dSi,dSp [B][HW].
 * gscale: device pointer to the upstream gradient of the jsd scalar (1 float). */
int onet_jsd_bwd(const float* gscale, const float* sums, const float* Si, int64_t Si_bs,
                 const float* Sp, int64_t Sp_bs, float* gL, float* dSi, float* dSp,
                 int B, int HW, void* stream);
/* elementwise quirk function, mutates x in place like OV:237-251; bwd: out = g * f'(x_original) */
int onet_log1pexp_inplace(float* x, int64_t n, void* stream);
int onet_log1pexp_bwd(const float* x, const float* g, float* out, int64_t n, void* stream);

/* ---- K11: argmax over the 2 classes, ties -> 0 (OV:201) ---------------------- */
int onet_argmax2(const float* S, int64_t* Y, int B, int HW, void* stream);

/* ---- "next" row f-3: the evaluation step either side of the path (UT = utils_20231218.py) ------- */
/* tensor_normal_per_frame (UT:673-689): y = (x - min) / (max - min + np.spacing(1)) per (b, c) plane */
int onet_normalise_per_frame(const float* x, float* y, int planes, int HW, void* stream);

/* ---- f-4: synthetic K-distributed sea clutter on the GPU: the frame recipe of the reference's generators (KD:469-526
 * `generate_K_distributed_noise`, KD:270-297, RG:177-216 `get_k_frame`, RG:63-175; NumPy statement: onet_amd/data.py), statistics
 * pinned to frames made by the reference's own functions (tests/golden/clutter_stats.npz).  `frames` frames of 400 x 400 (a
 * window of the 512 x 512 FFT torus) are synthesised: white Philox fields -> FFT colouring with the two complex frequency
 * responses filt_texture / filt_speckle ([512][512] float2 each, built by the host from the recipe: sqrt(fft2(R_G)) with R_G the
 * root field of the Hermite-coefficient polynomial, and sqrt(|f|^-0.6) on the recipe's index grid) -> MNLT to the Gamma(5)
 * texture x complex speckle; `n_targets` rotated Gaussian extended targets per frame are placed in order as
 * bg += (template > bg) * template (targets [frames][n_targets][8] = window x, window y, half-width, half-height, quadratic form
 * a, b, c, label threshold; peak amplitude sqrt(10^(snr_db[f] / 10) * mean clutter power)), and the centred H x W crop is
 * written to out [frames][H][W] (label [frames][H][W], may be NULL; fields [frames][4][400][400] = texture, Re speckle,
 * Im speckle, clutter amplitude, may be NULL).  Per-frame normalisation to [0,1] is onet_normalise_per_frame (UT:673-689).
 * Bit-reproducible in (seed, frame index): ranks generate disjoint frame ranges by passing different seeds. */
int onet_clutter_fft_size(void);
int onet_clutter_frame_size(void);
int64_t onet_clutter_ws_bytes(int frames);
int onet_clutter_generate(float* out, float* label, float* fields, const float* targets, const float* snr_db, int n_targets,
                          int frames, int H, int W, uint64_t seed, const float* filt_texture, const float* filt_speckle, void* ws,
                          int64_t ws_bytes, void* stream);
/* per-image 2-class confusion counts [B][4] = (TP, FP, FN, TN), positive class = 1: the inputs of
 * _acc/_miou/_target_iou/_detection_rate/_false_alarm_rate (UT:100-192) */
int onet_confusion2(const int64_t* pred, const int64_t* target, int64_t* counts, int B, int HW,
                    void* stream);
/* out = 1 - pred (hard label re-assignment, UT:429) */
int onet_flip_labels(const int64_t* pred, int64_t* out, int64_t n, void* stream);

/* ---- "next" row f-1: fused Adam over a flat buffer (TS:181-182) ---------------- */
/* torch.optim.Adam semantics (no amsgrad, L2 weight_decay): step is 1-based. */
int onet_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---- misc -------------------------------------------------------------------- */
int onet_fill(float* p, float value, int64_t n, void* stream);
/* y (+)= a*x, used for gradient accumulation of skip tensors */
int onet_axpy(const float* x, int64_t x_bs, float* y, int64_t y_bs, float a, int B, int64_t n,
              int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ONET_HIP_H */
