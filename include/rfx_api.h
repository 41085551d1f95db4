/*
 * rfx_api.h -- C ABI of librfx.so: the MI355X (gfx950 / CDNA4) kernels behind the RANSAC-Flow
 * coarse-to-fine alignment hot path.
 *
 * The reference (XiSHEN0220/RANSAC-Flow) is pure Python on PyTorch: it has no FFI of its own, its
 * "native" layer is the set of ATen/cuDNN/LAPACK ops its hot path issues.
 * Each entry point below replaces one of those op groups; the reference call site it stands in for
 * is cited as file:line relative to the reference root.  The Python drop-in modules
 * (ransac-flow_amd/dropin/{outil,model,coarseAlignFeatMatch}.py) keep the reference's Python
 * surface and bind these symbols through ctypes (see INTEGRATION.md for the stub a maintainer adds).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in
 *     _host; all tensors are dense, row-major, float32 ("f32") unless stated; NCHW images.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), allocates nothing, and never throws: the return value is 0 on success, a negative
 *     RFX_E_* code for a bad argument, or a positive hipError_t from the launch.
 *   - workspaces are caller-owned; rfx_*_ws_bytes() reports the size needed.
 */
#ifndef RFX_API_H
#define RFX_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RFX_OK 0
#define RFX_E_ARG (-1)     /* null pointer / non-positive size / unsupported geometry */
#define RFX_E_LIMIT (-2)   /* size above a compiled-in limit */

#define RFX_ACT_NONE 0
#define RFX_ACT_RELU 1
#define RFX_ACT_SIGMOID 2

/* library / build identification; returns a static string "rfx <version> gfx950". */
const char* rfx_version(void);

/* ABI revision of this header: bumped whenever an entry point changes its signature or its operand layout (round 2:
 * rfx_compose_flow_f32 gained Hc/Wc, 3x3/s1/p1 geometries moved to rfx_conv3x3_f32's packed weights; round 3: multi-homography
 * round kernels, two-direction correlation, grouped launches; round 4: rfx_draw_samples_i64 keyed by pair id).  A binding
 * compares rfx_abi_version() with the RFX_ABI_VERSION it was written against and refuses a mismatch. */
#define RFX_ABI_VERSION 10
int rfx_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Grouped launches: ONE launch per kernel instance for the same layer of several INDEPENDENT problems of different
 * sizes.  A single pair runs the trunk on 8 images of 8 sizes (7 pyramid levels + the target,
 * quick_start/coarseAlignFeatMatch.py:92-125): layer by layer that is 8 launches of 2-150 workgroups on 256 CUs, bound
 * by one workgroup lifetime per layer AND level.  Between rfx_group_begin() and rfx_group_end() the convolution entry
 * points (rfx_conv2d_f32, rfx_conv3x3_f32, rfx_conv3x3_conv1x1_f32, rfx_stem_conv7x7_maxpool_f32) validate and RECORD
 * their launch instead of issuing it; rfx_group_end(stream) issues, per kernel instance, one launch whose blockIdx.y
 * selects the problem (up to 8 per launch).  The device code of a problem is the single launch's, so results are
 * bit-identical.  Recording is per host thread, groups do not nest, every other entry point launches immediately;
 * the caller keeps all operands alive until rfx_group_end and records only mutually independent calls.
 * rfx_group_abort() drops a recording without launching.  The kernel instances of one group that remain distinct run
 * concurrently on two library-owned side streams forked from / joined to `stream`; rfx_group_side_streams(0) turns that off
 * for the calling host thread (returns the previous setting) -- a caller that already runs several grouped chains on streams
 * of its own (rfx/pipeline.py: the images of a single pair as two chains) wants each chain serial on its stream.
 * ------------------------------------------------------------------------------------------ */
int rfx_group_begin(void);
int rfx_group_end(void* stream);
int rfx_group_abort(void);
int rfx_group_side_streams(int enable);

/* ------------------------------------------------------------------------------------------
 * Convolution family (ResNet-50 conv1..layer3 trunk: model/resnet50.py:68-104,112-169 as used by
 * quick_start/coarseAlignFeatMatch.py:34-52,106,124; FeatureExtractor model/model.py:59-125;
 * NetFlowCoarse / NetMatchability convs model/model.py:210-226,289-305).
 *
 * rfx_conv2d_f32: implicit-GEMM convolution on the fp32 MFMA (v_mfma_f32_32x32x2_f32), NCHW,
 * groups=1, dilation=1, with a fused per-output-channel affine (eval-mode BatchNorm folded to
 * scale/shift), optional residual add and activation:
 *     out[n,m,oh,ow] = act( scale[m] * sum_{c,kh,kw} in[n,c,oh*s-p+kh,ow*s-p+kw] * w[m,c,kh,kw]
 *                           + shift[m] + residual[n,m,oh,ow] )
 * wT    : packed weights [Kpad][Mpad], wT[k*Mpad+m] = w[m,c,kh,kw] with k=(c*KH+kh)*KW+kw,
 *         Kpad = roundup(Cin*KH*KW,32), Mpad = roundup(Cout,128), zero padded.
 * ktab  : int32[Kpad], (c<<8)|(kh<<4)|kw for k < K and -1 for the padding rows, stored per block of 32
 *         consecutive k as [16 even k | 16 odd k].
 * scale/shift may be NULL (treated as 1 / 0); residual may be NULL.
 * ------------------------------------------------------------------------------------------ */
int rfx_conv2d_f32(const float* in, const float* wT, const int32_t* ktab, const float* scale,
                   const float* shift, const float* residual, float* out, int N, int Cin, int Hin,
                   int Win, int Cout, int KH, int KW, int stride, int pad, int act, void* stream);

/* Which tile configuration rfx_conv2d_f32 launches for an output of (N, Cout, Hout, Wout):
 * 0 = 128x128 (kernel conv2d_mfma_kernel<2,2>), 1 = 64x128 (<1,2>), 2 = 64x64 (<1,1>).  Lets a profiler-side
 * caller attribute algorithmic FLOPs to the kernel instance rocprofv3 reports. */
int rfx_conv2d_tile_variant(int N, int Cout, int Hout, int Wout);

/* Full kernel-instance id a convolution of this geometry runs as: bits 0-1 = tile variant above, bit 2 = 1x1 specialisation, bit 3 =
 * wave-specialised form (4 MFMA + 4 loader wavefronts), bit 4 = 16-byte pixel-side loads (1x1, stride 1):
 * the template arguments <TM,TN,ONE,WS,VECB> rocprofv3 prints.  Bit 5 = the direct 3x3 / stride 1 / pad 1 kernel
 * conv3x3_direct_kernel<TM, PT_C> (Cin >= 8), TM = 2 if bits 0-1 are 0 else 1; bits 6-7 = output patch shape
 * (0: 8x16, 1: 16x8, 2: 32x4 -- the one that pads the H x W map least); bit 11 = 256-pixel (16x16) patches, conv3x3_direct_kernel<1, 16, false, 4>
 * (Cout <= 64 on launches of >= 1024 such patches: twice the pixels per staged weight image); bit 12 = the instance with a ragged last K
 * step (Cin % 8 != 0), conv3x3_direct_kernel<TM, PT_C, false, 2, true>.  Bit 10 = the k-major 1x1 / stride 1 kernel
 * conv1x1_kmajor_kernel<TM, VEC> (Cin % 32 == 0; TM = 2 - bits 0-1, VEC = bit 4).  Bit 5 set = the geometry is served by
 * rfx_conv3x3_f32 below (the host mirrors call it then); rfx_conv2d_f32 itself always runs the implicit-GEMM kernel. */
int rfx_conv2d_kernel_id(int N, int Cin, int Cout, int KH, int KW, int stride, int pad, int Hout, int Wout);
/* (round 4) bit 13 = the direct 3x3 / STRIDE 2 / pad 1 kernel conv3x3_s2_kernel<TM> (Cin % 8 == 0, TM = 2 - bit 0), served by
 * rfx_conv3x3_s2_f32 below: ResNet-50 layer2.0 / layer3.0 conv2 (model/resnet50.py:75) and the first convolution of
 * FeatureExtractor layer2 / layer3 (model/model.py:86-95).
 * bit 14 = chunked accumulation: on layers whose K is long -- 3x3 / stride 1 with K = 9 Cin >= 2048 (conv3x3_direct_kernel<TM, PT_C,
 * false, 2, false, 4>, a chunk = 4 K steps = 288 products) and 1x1 / stride 1 with K >= 1024 (conv1x1_kmajor_kernel<1, VEC, 8>, a chunk
 * = 256 products, 64-channel tiles at every launch size) -- the accumulators are added to a running total at the end of every chunk and
 * restarted, the K-blocked sum of the reference's CPU kernels (MKL sgemm / oneDNN) instead of ONE fma chain over K: 2.7-3.8x less
 * round-off against float64 at K = 2304 / 4608.  On those 3x3 layers rfx_conv2d_f32 (its implicit-GEMM kernel: a chain) and
 * rfx_conv3x3_f32 therefore differ in the last bits -- everywhere else they are bit-identical; the 1x1 layers with K >= 1024 run chunked
 * through rfx_conv2d_f32 itself.  RFX_C3_CHUNK=0 / RFX_C1_CHUNK=0: chains (A/B runs). */

/* 3x3 / stride 1 / pad 1 convolution, Cin >= 8 (a Cin that is not a multiple of 8 -- the 49-channel correlation volume in
 * front of the heads -- takes ceil(Cin/8) K steps, the packed weights carrying zero rows for the missing channels) (ResNet Bottleneck conv2 at stride 1, model/resnet50.py:75; the
 * FeatureExtractor BasicBlocks, model/model.py:32-35; the NetFlowCoarse / NetMatchability stacks, model/model.py:170-181):
 * the direct kernel that stages the raw input patch in LDS.  Same epilogue and the same result, bit for bit, as
 * rfx_conv2d_f32; the weights come packed in the kernel's own LDS order so that staging is a straight copy:
 *     wP[mt][s][h][m][kk] = w[mt*128 + m, c, kh, kw]   with k = (c*3 + kh)*3 + kw = s*72 + 2*kk + h,
 *     mt < roundup(Cout,128)/128, s < ceil(Cin/8), h < 2, m < 128, kk < 36; zero for mt*128 + m >= Cout and for c >= Cin; 16-byte aligned.
 * k_chunk (ABI 8): 0 = the library's rule (chunked accumulation, bit 14 above, for K = 9 Cin >= 2048); 4 = close a chunk every 4 K
 * steps (288 products) whatever K is -- what rfx_conv3x3_conv1x1_f32 does in its 3x3 phase since round 5, so that a Bottleneck
 * tail run as two kernels (this one with k_chunk = 4, then the 1x1) equals the fused kernel bit for bit.  Other values: RFX_E_ARG.
 * rfx_conv3x3_kernel_id: the instance this call launches (bits as rfx_conv2d_kernel_id). */
int rfx_conv3x3_f32(const float* in, const float* wP, const float* scale, const float* shift, const float* residual,
                    float* out, int N, int Cin, int H, int W, int Cout, int act, int k_chunk, void* stream);
int rfx_conv3x3_kernel_id(int N, int Cin, int Cout, int H, int W, int k_chunk);
/* The same for stride 2 (pad 1, Cin % 8 == 0): out (N, Cout, (H-1)/2+1, (W-1)/2+1); wP as above.  Bit-identical to
 * rfx_conv2d_f32 on the same geometry for K = 9 Cin < 1152 ONLY: from Cin = 128 on (K >= 1152) this kernel sums in chunks of 288
 * products (conv3x3_s2_kernel<TM, 4>: bit 14 of the kernel id), the implicit-GEMM kernel behind rfx_conv2d_f32 in ONE chain -- the two
 * then differ in the last bit, and so do trunk features when the host mirror is switched to the generic kernel (RFX_CONV_S2=0 /
 * RFX_CONV_DIRECT=0 are A/B switches for experiments, not product settings: they change the sums). */
int rfx_conv3x3_s2_f32(const float* in, const float* wP, const float* scale, const float* shift, const float* residual,
                       float* out, int N, int Cin, int H, int W, int Cout, int act, void* stream);

/* Bottleneck tail (model/resnet50.py:71-79, forward :93-103: conv2 3x3 -> bn2 -> relu -> conv3 1x1 -> bn3 -> += residual
 * -> relu) as ONE kernel: out = act3(bn3(conv1x1(act2(bn2(conv3x3(in))))) + residual).  The workgroup that computed a
 * 128-pixel tile of ALL Cmid channels of the 3x3 convolution keeps it in LDS and multiplies it with the expansion weights,
 * so the Cmid-channel map never goes to HBM.  in (N,Cin,H,W); wP2/scale2/shift2 = weights (packed as for
 * rfx_conv3x3_f32) / folded BN of the 3x3 (stride 1, pad 1, Cin % 8 == 0, Cmid = 64 or 128); scale3 / shift3 of the 1x1 (Cexp % 128 == 0) and its weights in
 * "quad" order wQ3[q][h][m][j] = W3[m][8q + 2j + h] (q < Cmid/8, h < 2, m < Cexp, j < 4; 16-byte aligned): the four MFMA A
 * operands of a lane for four consecutive k-pairs are one 16-byte load, coalesced over the channels;
 * residual (N,Cexp,H,W) or NULL; act2 / act3 = RFX_ACT_NONE or RFX_ACT_RELU.
 * Round 5: the 3x3 phase (K = 9 Cmid = 576 / 1152) closes a chunk every 4 K steps (288 products) into a second accumulator set --
 * the last place the device summed a long K as ONE fma chain while the reference's oneDNN kernels block it.
 * Bit-identical to rfx_conv3x3_f32(k_chunk = 4) followed by the 1x1 through rfx_conv2d_f32.  RFX_C3_TAIL_CHUNK=0: round 4's chain. */
/* Kernel instance of the launch below (conv3x3_direct_kernel<TM, PT_C, true, 2, false, KCH> as rocprofv3 prints it): bit 9 set, bit 0 =
 * 64-channel mid tile (TM = 1), bits 6-7 = output patch shape as in rfx_conv2d_kernel_id, bit 14 = chunked (KCH = 4). */
int rfx_conv3x3_conv1x1_kernel_id(int N, int H, int W, int Cmid);
int rfx_conv3x3_conv1x1_f32(const float* in, const float* wP2, const float* scale2, const float* shift2, int act2,
                            const float* wQ3, const float* scale3, const float* shift3, const float* residual, int act3,
                            float* out, int N, int Cin, int H, int W, int Cmid, int Cexp, void* stream);

/* nn.MaxPool2d(k, stride, pad) with -inf padding (model/resnet50.py:120: k=3,s=2,p=1;
 * model/model.py:71: k=2,s=1,p=0).  Hout = (Hin+2p-k)/s+1. */
int rfx_maxpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int k, int stride,
                      int pad, void* stream);

/* Anti-aliased down-sampling (model/downsample.py:12-46, filt_size=3): reflect-pad 1, depthwise
 * [1 2 1]x[1 2 1]/16, stride.  Hout = (Hin-1)/stride+1. */
int rfx_blurpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int stride, void* stream);

/* Fused MaxPool2d(kernel 2, stride 1) + BlurPool(stride): the FeatureExtractor stem `self.maxpool`
 * (model/model.py:71-72).  Hout = (Hin-2)/stride+1.  Bit-identical to the two separate calls. */
int rfx_maxblurpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int stride, void* stream);

/* The whole FeatureExtractor stem in one kernel (model/model.py:68-72, forward :106-110): conv3x3(3 -> Cout, stride 1,
 * pad 1) + folded BatchNorm + ReLU + MaxPool2d(2, stride 1) + BlurPool(stride 2).  in (N,3,H,W); wT / scale / shift as
 * for rfx_conv2d_f32 (the packed weights of that convolution: K = 27 -> Kpad = 32 rows); out (N,Cout,(H-2)/2+1,(W-2)/2+1).
 * Bit-identical to rfx_conv2d_f32(act = ReLU) followed by rfx_maxblurpool2d_f32(stride 2); the full-resolution
 * Cout-channel map never goes to HBM.  Cout % 32 == 0. */
int rfx_stem_conv3x3_maxblur_f32(const float* in, const float* wT, const float* scale, const float* shift, float* out,
                                 int N, int H, int W, int Cout, void* stream);

/* The ResNet-50 stem in one kernel (model/resnet50.py:115-120, forward :157-160): conv7x7(3 -> Cout, stride 2, pad 3) +
 * folded BatchNorm + ReLU + MaxPool2d(3, stride 2, pad 1).  in (N,3,H,W); wT / scale / shift = the packed weights of that
 * convolution (K = 147 -> Kpad = 160 rows); out (N,Cout,Hp,Wp) with Hc = (H-1)/2+1, Hp = (Hc-1)/2+1 (same for W).
 * Bit-identical to rfx_conv2d_f32(act = ReLU) followed by rfx_maxpool2d_f32(3, 2, 1).  Cout % 32 == 0. */
int rfx_stem_conv7x7_maxpool_f32(const float* in, const float* wT, const float* scale, const float* shift, float* out,
                                 int N, int H, int W, int Cout, void* stream);

/* F.normalize(x, p=2, dim=1, eps=1e-12) on NCHW (quick_start/coarseAlignFeatMatch.py:106,124;
 * quick_start/align2images.py:87-88).  `in` is dense; element (n,c,p) of the result goes to
 * out[n*out_batch_stride + c*out_chan_stride + p] (0 = dense defaults C*HW / HW), which lets the coarse
 * aligner write every pyramid scale straight into the concatenated (C, nA) match matrix
 * (quick_start/coarseAlignFeatMatch.py:108,115).  In-place allowed for the dense case. */
int rfx_l2norm_nchw_f32(const float* in, float* out, int N, int C, int HW, long long out_batch_stride,
                        long long out_chan_stride, void* stream);

/* NetFlowCoarse tail (model/model.py:228-233): softmax over the K*K logits, expectation of the tap
 * offsets; flow[n,0] = sum_q p_q * gx_q / cols * 2, flow[n,1] = sum_q p_q * gy_q / rows * 2 with
 * q = i*K+j, gx_q = j-K/2, gy_q = i-K/2.  logits (N,K*K,rows,cols) -> flow (N,2,rows,cols). */
int rfx_flow_head_f32(const float* logits, float* flow, int N, int K, int rows, int cols, void* stream);

/* Bilinear resize of NCHW maps: F.interpolate(mode='bilinear', align_corners=False)
 * (quick_start/align2images.py:92, evaluation/evalHpatch/evaluation.py:37-40) or
 * F.upsample_bilinear (= align_corners=True; model/model.py:234,309). */
int rfx_resize_bilinear_f32(const float* in, float* out, int NC, int Hin, int Win, int Hout, int Wout,
                            int align_corners, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image pre-processing on the device (host-side in the reference:
 * quick_start/coarseAlignFeatMatch.py:55-57,80-110).
 * ------------------------------------------------------------------------------------------ */
/* One separable pass of Pillow's fixed-point LANCZOS resampler on uint8 NHWC images (C <= 4).
 * vertical = 0: out (N, outH, outW, C) with out row y filtered from source row y + row_offset (outH + row_offset
 * <= inH); vertical = 1: out (N, outH, inW, C) (outW must equal inW).  bounds: int32 (n_out, 2) = [first source
 * index, tap count] per output coordinate; weights: int32 (n_out, ksize) = round(w * 2^22) (Pillow's
 * precompute_coeffs + normalize_coeffs_8bpc; rfx/lanczos.py computes them).  Integer arithmetic: bit-identical to
 * PIL.Image.resize(resample=LANCZOS) when the two passes are chained as Pillow does. */
int rfx_lanczos_pass_u8(const uint8_t* in, uint8_t* out, int N, int inH, int inW, int outH, int outW, int C,
                        const int32_t* bounds, const int32_t* weights, int ksize, int vertical, int row_offset,
                        void* stream);

/* torchvision ToTensor (+ Normalize): uint8 (N,H,W,3) -> float32 (N,3,H,W).  raw = x/255 (may be NULL),
 * norm = (x/255 - mean[c]) / std[c] (may be NULL); mean3_host / std3_host are HOST pointers to 3 floats. */
int rfx_u8_to_f32_chw(const uint8_t* in, float* raw, float* norm, int N, int H, int W, const float* mean3_host,
                      const float* std3_host, void* stream);

/* Row-pitch change of a (rows, w_src) float32 matrix into (rows, w_dst): copies min(w_src, w_dst) columns per row and
 * zero-fills the rest.  Used to run the correlation on zero-padded copies of maps whose width is not a multiple of 4 (the
 * half-resolution pass of the KITTI driver, evaluation/evalKITTI/evaluation.py:290: 1072/8 = 134 columns) and to crop its
 * output; zero padding on the right is exactly CorrNeigh's own padding (model/model.py:135,143), so results are unchanged. */
int rfx_copy_cols_f32(const float* src, float* dst, long long rows, int w_src, int w_dst, void* stream);

/* ------------------------------------------------------------------------------------------
 * 7x7 local correlation volume (model/model.py:129-160, CorrNeigh.do_forward):
 *     out[n, i*K+j, r, c] = sum_ch x[n,ch,r,c] * y[n,ch,r+i-K/2,c+j-K/2]   (zero outside y)
 * K must be 7 (the only size the reference instantiates: quick_start/align2images.py:38).
 * C must be a multiple of 8.
 * ------------------------------------------------------------------------------------------ */
int rfx_corr_neigh_f32(const float* x, const float* y, float* out, int N, int C, int H, int W, int K,
                       void* stream);
/* Same operation with the kernel configuration forced (tuning / tests / roofline decomposition; variant 0 = the
 * automatic choice of rfx_corr_neigh_f32): 1..3 = 64/32/16-row x 16-column tiles, 4 = 16 rows x 80 columns (whole
 * 480x640-pair feature-map width) plain, 5 = the tuned 16x80 kernel (3 tap groups, hand-pipelined LDS reads, balanced wave
 * map, masked DMA, equal row tiles), 6 = 5 with 2 tap groups, 7 / 8 = the tuned kernel with 48- / 64-column tiles, 9 = 32x32.  Variants 1..9 give
 * bit-identical results (channel-ordered fmaf sums).  Unknown variant -> RFX_E_ARG.  (The roofline-decomposition variants 21..24
 * -- compute / DMA / LDS reads / FMAs removed, WRONG results -- exist only in experiment builds, -DRFX_CORR_EXPERIMENTS:
 * `make exp NAME=correxp SRC=corr DEFS=-DRFX_CORR_EXPERIMENTS`; the product library rejects them.) */
int rfx_corr_neigh_variant_f32(const float* x, const float* y, float* out, int N, int C, int H, int W, int K,
                               int variant, void* stream);

/* Both directions of a pair in ONE pass over the features -- what PredFlowMask computes with two CorrNeigh calls
 * (evaluation/evalHpatch/evaluation.py:29,34: corr12 = netCorr(featt, featsSample), corr21 = netCorr(featsSample, featt)):
 *     out_xy = CorrNeigh(x, y),   out_yx = CorrNeigh(y, x),   each (N, K*K, H, W).
 * The reverse volume is the forward one at mirrored taps and shifted pixels,
 *     out_yx[n, (K-1-i)*K + (K-1-j), r+i-K/2, c+j-K/2] = out_xy[n, i*K+j, r, c]
 * (the same channel-ordered products, so the values are bit-identical to a second rfx_corr_neigh_f32 call), and zero where
 * the source pixel lies outside the image: the kernel stores every accumulator twice instead of reading 2*C*H*W floats a
 * second time -- (2C + 2K^2)*4 bytes per pixel instead of 2*(2C + K^2)*4.  Requires W % 4 == 0, C % 2 == 0 and 16-byte aligned
 * pointers (the host mirror pads other widths with rfx_copy_cols_f32); K must be 7. */
int rfx_corr_neigh_bidir_f32(const float* x, const float* y, float* out_xy, float* out_yx, int N, int C, int H, int W, int K,
                             void* stream);
/* Kernel-duration capture for the roofline of THIS kernel (bench.py): while a host thread has it enabled, every correlation launch of
 * that thread is issued with hipExtLaunchKernelGGL and a library-owned start / stop event pair attached to the dispatch itself (the
 * timestamps rocprofv3 reads; an event pair recorded around a launch also brackets ~18 us of command-processor work).
 * rfx_corr_timing(enable) returns the previous setting; rfx_corr_timing_collect waits for the captured launches, writes their
 * durations in microseconds in launch order (up to cap; us_out may be NULL), releases the events and returns how many there were. */
int rfx_corr_timing(int enable);
int rfx_corr_timing_collect(float* us_out, int cap);

/* ------------------------------------------------------------------------------------------
 * Warping (kornia HomographyWarper.warp_grid: quick_start/align2images.py:61,65;
 * F.grid_sample: quick_start/align2images.py:66,95,97, evaluation/evalHpatch/evaluation.py:25,45).
 * ------------------------------------------------------------------------------------------ */
/* grid[b,y,x,:] = proj( Hm[b] * (lin(x,w), lin(y,h), 1) ),  lin(i,n) = -1 + 2 i/(n-1). */
int rfx_warp_grid_f32(const float* Hm, float* grid, int B, int h, int w, void* stream);

/* bilinear, zeros padding; input (N,C,Hi,Wi), grid (N,Ho,Wo,2) -> out (N,C,Ho,Wo). */
int rfx_grid_sample_f32(const float* in, const float* grid, float* out, int N, int C, int Hi, int Wi,
                        int Ho, int Wo, int align_corners, void* stream);

/* Fused fine-flow composition (quick_start/align2images.py:92-95;
 * evaluation/evalHpatch/evaluation.py:40-45,51): up-sample flowDown (N,2,hd,wd) to (H,W) with
 * align_corners=False, add the linspace identity grid, optionally clamp to [-1,1], sample the coarse
 * grid (N,Hc,Wc,2) there (bilinear, zeros, align_corners=False) -> flow12 (N,H,W,2).  (Hc,Wc) = (H,W) everywhere except in
 * the full-resolution pass of the KITTI driver, which composes at the ORIGINAL image size on a coarse grid of the
 * fine-stage size (evaluation/evalKITTI/evaluation.py:302 with :49-81).  If inb != NULL it
 * receives the in-bounds mask (N,H,W) = (-1<=fx<=1)&(-1<=fy<=1) as 0/1 floats; if flowUp != NULL it
 * receives the (clamped) sampling grid (N,H,W,2). */
int rfx_compose_flow_f32(const float* flowDown, const float* coarseGrid, float* flow12, float* inb,
                         float* flowUp, int N, int hd, int wd, int Hc, int Wc, int H, int W, int clamp, void* stream);

/* The tail of model.predFlowCoarse / predFlowCoarseNoGrad (model/model.py:333-340, 344-350): flowCoarse (B,2,H,W) is the
 * NetFlowCoarse output; flow (B,H,W,2) = clamp(flowCoarse.permute(0,2,3,1) + grid, -1, 1) with grid (grid_batch,H,W,2),
 * grid_batch = 1 (broadcast) or B; flowGrad (B,1,H-1,W-1) = L2 norm over the two channels of
 * flowCoarse[:, :, 1:, 1:] - flowCoarse[:, :, :-1, :-1], or NULL (the NoGrad variant). */
int rfx_flow_grad_clamp_f32(const float* flowCoarse, const float* grid, float* flowGrad, float* flow, int B, int H, int W,
                            int grid_batch, void* stream);

/* Multi-homography merge of the offline flow assembly (evaluation/evalHpatch/getResults.py:48-61,
 * evaluation/evalCorr/getResults.py:121-134, evaluation/evalKITTI/getResults.py:126-138).
 * flow (n,HW,2) composed flows; match12 = up-sampled matchability of homography i at match12 + i*match12_stride;
 * cyc (n,HW) = grid_sample(up(match21), flowUp) or NULL; inb (n,HW) in-bounds mask or NULL.
 * score_i = match12_i * cyc_i * inb_i (left to right).  Pixel owner: homography 0 if score_0 >= th, otherwise the
 * first i >= 1 with score_i >= th (only when multiH != 0), otherwise 0.  Outputs: flowGlobal (HW,2) = owner's flow
 * clamped to [-1,1]; matchGlobal (HW) = owner's score (may be NULL); binary (HW) u8 = "some score reached th"
 * (may be NULL). */
int rfx_merge_multi_h_f32(const float* flow, const float* match12, long long match12_stride, const float* cyc,
                          const float* inb, int n, long long HW, float th, int multiH, float* flowGlobal,
                          float* matchGlobal, uint8_t* binary, void* stream);

/* score (n,HW) = match12_i * cyc_i * inb_i (same operands and order as above; cyc / inb may be NULL): the "match"
 * tensor of evaluation/evalKITTI/getResults.py:120, materialised when a host-side filter (remove_small_cc, :122)
 * runs before the merge -- rfx_merge_multi_h_f32 is then called with cyc = inb = NULL on the filtered scores. */
int rfx_match_score_f32(const float* match12, long long match12_stride, const float* cyc, const float* inb, int n,
                        long long HW, float* score, void* stream);

/* The small-component filter of the KITTI scripts (evaluation/evalKITTI/evaluation.py:85-100 online -- remove_small_cc(match,
 * 0.99, cc_th) at :321 -- and evaluation/evalKITTI/getResults.py:66-84 offline, :122), which the reference runs on the host
 * with skimage.measure.label: out = in with every 8-connected component of (in > match_th) of at most max_area pixels set to
 * zero, per image of the (N,H,W) batch.  max_area = the largest pixel count a with a / (H*W) <= cc_th in float64 (the host
 * mirrors compute it so; the reference's test is on the area FRACTION).  ws: rfx_remove_small_cc_ws_bytes(N,H,W) bytes.
 * Exact: the result of a labelling does not depend on the label numbering.  in and out may be the same buffer. */
size_t rfx_remove_small_cc_ws_bytes(int N, int H, int W);
int rfx_remove_small_cc_f32(const float* in, float* out, int N, int H, int W, float match_th, int max_area, void* ws,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * All-pairs correlation + mutual nearest neighbours (utils/outil.py:32-45, mutualMatching).
 * featA: (C, nA) and featB: (C, nB), column = one cell ("K-major": element (k,i) at k*ld+i).
 * maskB (optional, nB floats, 0/1) multiplies featB's columns (quick_start/coarseAlignFeatMatch.py:143).
 * Outputs: idx1/idx2 (int64, capacity min(nA,nB)) in ascending idx1 order, count (int32[1]).
 * The nA x nB score matrix is never written to memory.  A score -- a sum of C non-negative products -- is accumulated in chunks of
 * 256 products (default; argument score_chunk) that are added to a running total (round 4): the chain's round-off over C = 1024 is
 * 2.8x that of the reference's torch.mm and flipped float64 near-ties of the arg-max 1.5-2.3x as often as the reference flips them
 * against itself; chunked: 1.0-1.1x on the 64 bench pairs, 1.25x over 128 / 160 pairs.
 * ws: rfx_mutual_nn_ws_bytes(nA, nB) bytes.
 * ------------------------------------------------------------------------------------------ */
size_t rfx_mutual_nn_ws_bytes(int nA, int nB);
/* (ABI 8) score_chunk: how a score's C products are summed -- an fma chain inside chunks of `score_chunk` consecutive products
 * (a multiple of 32), the chunk sums added to a running total in k order.  0 = the library default, 256 products; < 0 = ONE chain
 * over all C.  The reference's score is torch.mm on the HOST (utils/outil.py:34): MKL's sgemm sums k as an fma chain inside blocks
 * of KC products and adds the block sums to C -- KC = 192 on the GPU box's EPYC host, 384 on a Xeon (scripts/mm_blocking_probe.py).
 * With score_chunk = KC the device's scores equal that host's torch.mm BIT FOR BIT on equal features, so that what is left of the
 * arg-max near-tie flips comes from the trunk features alone.  It is an ARGUMENT of every call (ABI 7 had a process-wide setter):
 * two pipelines of one process, or two ranks, never share it by accident; the Python mirror resolves it ONCE per pipeline
 * (rfx/ops.py::resolve_score_chunk: an explicit value, or a probe of the host's KC) and records it in its results.
 * Returns RFX_E_ARG for a positive score_chunk that is not a multiple of 32. */
int rfx_mutual_nn_f32(const float* featA, int ldA, int nA, const float* featB, int ldB, int nB, int C,
                      const float* maskB, int64_t* idx1, int64_t* idx2, int32_t* count, void* ws, int score_chunk,
                      void* stream);

/* The same for `batch` independent pairs in one launch (blockIdx.y = pair): pair b reads featA + b*strideA,
 * featB + b*strideB, maskB + b*nB, writes idx1/idx2 + b*min(nA,nB), count[b], and uses ws + b*ws_bytes(nA,nB). */
int rfx_mutual_nn_batched_f32(const float* featA, int ldA, int nA, long long strideA, const float* featB, int ldB,
                              int nB, long long strideB, int C, const float* maskB, int64_t* idx1, int64_t* idx2,
                              int32_t* count, void* ws, int batch, int score_chunk, void* stream);

/* ------------------------------------------------------------------------------------------
 * RANSAC over 4-point DLT homographies (utils/outil.py:68-164).
 * match1/match2: (n,3) source / target points (x,y,1); samples: (N,4) int64 indices into them.
 * ------------------------------------------------------------------------------------------ */
/* outil.Homography (utils/outil.py:68-87): per hypothesis the unit-norm null vector of the 8x9 DLT
 * system in float64 with LAPACK dgesdd's sign (dgebd2 Householder sweep), cast to float32.
 * X,Y: (N,4,3) source / target samples -> Hout (N,3,3). */
int rfx_dlt4_homography(const float* X, const float* Y, int N, float* Hout, void* stream);

/* outil.Prediction (utils/outil.py:97-100): reprojection error ||X[:, :2] - proj(Y H^T)||_2 for every
 * (hypothesis, match) pair, float32 in the reference's operation order.  match1/match2 (n,3), Hs (N,3,3)
 * -> err (N,n).  N <= 65535. */
int rfx_prediction_f32(const float* match1, const float* match2, int n, const float* Hs, int N, float* err,
                       void* stream);

/* outil.ScoreRANSAC (utils/outil.py:102-113): H per hypothesis + inlier count * (det(H) > 1e-6).
 * Hout (N,3,3) f32, counts (N) int64.  ws: rfx_ransac_ws_bytes(n, N).
 * flags_out (ABI 8; optional, N bytes): the per-hypothesis flags of the SAME DLT pass -- bit 0 evaluated, bit 1 det gate passed,
 * bit 2 the 8x9 system is rank deficient (dlt.h: one Sturm count on the bidiagonal) -- for a caller that re-solves exactly those
 * hypotheses with the host's LAPACK (rfx/ops.py::score_hypotheses(degenerate="lapack")); NULL skips the rank test. */
int rfx_score_hypotheses(const float* match1, const float* match2, int n, const int64_t* samples, int N,
                         float tol, float* Hout, int64_t* counts, uint8_t* flags_out, void* ws, void* stream);

/* outil.RANSAC (utils/outil.py:117-164) given the index draw: duplicate filter (:122-133), chunks of
 * 100 with the zero-chunk abort (:140-146), tail chunk (:153-160), first-maximum selection, final
 * inlier mask (:162-163).
 * result (int32[4]): [0] status 0 = ok, 1 = aborted (reference returns (None,0,[],[])), 2 = no
 * hypothesis beat 0 (reference raises TypeError); [1] best inlier count; [2] index of the winning
 * hypothesis in `samples`; [3] number of hypotheses that survived the duplicate filter.
 * bestH (9 f32), inlier (n uint8).  ws: rfx_ransac_ws_bytes(n, N). */
size_t rfx_ransac_ws_bytes(int n, int N);
int rfx_ransac_h4(const float* match1, const float* match2, int n, const int64_t* samples, int N,
                  float tol, float* bestH, uint8_t* inlier, int32_t* result, void* ws, void* stream);

/* The same for a whole batch of pairs in ONE launch chain (pack, DLT, count, select with blockIdx.y = pair): what the
 * per-pair host loop around outil.RANSAC (quick_start/coarseAlignFeatMatch.py:157-170, once per pair) costs in launches
 * and host round trips.  match1/match2 (batch,cap,3); n (batch) int32 ON THE DEVICE = matches of each pair (<= cap);
 * samples (batch,N,4) int64; bestH (batch,9); inlier (batch,cap) uint8; result (batch,4) as above with the extra
 * status 3 = fewer than nbPoint (4) matches (the reference returns its None sentinel before calling RANSAC,
 * quick_start/coarseAlignFeatMatch.py:157-158).  Per pair bit-identical to rfx_ransac_h4.  batch <= 65535.
 * ws: rfx_ransac_batched_ws_bytes(cap, N, batch). */
size_t rfx_ransac_batched_ws_bytes(int cap, int N, int batch);
int rfx_ransac_h4_batched(const float* match1, const float* match2, const int32_t* n, int cap, const int64_t* samples,
                          int N, float tol, float* bestH, uint8_t* inlier, int32_t* result, void* ws, int batch,
                          void* stream);

/* Rank-deficient 4-point samples (utils/outil.py:84: ``np.linalg.svd(A)`` on an 8x9 system of rank 7).  Three matched points
 * collinear in BOTH images -- common on the feature-cell lattices -- leave a two-dimensional null space; which unit vector of
 * it LAPACK's Vh[8] is depends on rounding inside the host BLAS (it differs between LAPACK builds), so no device restatement
 * can be bit-exact there.  The DLT kernel therefore FLAGS such hypotheses (sigma_8 < 1e-8 * max|bidiagonal|, one Sturm count on
 * the bidiagonal dgebd2 leaves; csrc/dlt.h) and the search can be run in two stages around a host re-solve:
 *   rfx_ransac_h4_batched_stage(..., stages = 1)   pack + DLT: hypotheses and flags into ws (bestH / inlier / result untouched)
 *   rfx_ransac_degenerate_list(ws, ...)            idx (batch,kcap) int32 = flagged hypotheses that survived the duplicate
 *                                                  filter, ascending; count (batch) int32 (may exceed kcap: list truncated)
 *   [host: numpy's LAPACK on exactly those systems -- what the reference itself computes on this host]
 *   rfx_ransac_patch_h(ws, ..., idx, count, Hpatch (batch,kcap,9))   re-solved homographies in, det gate re-evaluated
 *   rfx_ransac_h4_batched_stage(..., stages = 2)   count + select on the patched hypotheses
 * stages = 3 is rfx_ransac_h4_batched.  cap / N / batch / ws as for rfx_ransac_h4_batched (same workspace across the calls). */
int rfx_ransac_h4_batched_stage(const float* match1, const float* match2, const int32_t* n, int cap, const int64_t* samples,
                                int N, float tol, float* bestH, uint8_t* inlier, int32_t* result, void* ws, int batch,
                                int stages, void* stream);
int rfx_ransac_degenerate_list(const void* ws, int cap, int N, int batch, int32_t* idx, int32_t* count, int kcap, void* stream);
int rfx_ransac_patch_h(void* ws, int cap, int N, int batch, const int32_t* idx, const int32_t* count, const float* Hpatch,
                       int kcap, void* stream);
/* The same exchange in COMPACT form (round 6: what the exact mode runs on).  rfx_ransac_degenerate_gather, after stage 1: lists the
 * flagged hypotheses of every pair (idx (batch,N) int32, count (batch) int32 -- device scratch), off (batch+1) int32 = exclusive
 * prefix of count with off[batch] = total (device; also stored to off_host when non-NULL), and writes ONE ROW PER FLAGGED
 * HYPOTHESIS at row off[b]+k of xy_rows: 16 floats = the sample's 4 source points (x, y) then its 4 target points (x, y), i.e.
 * X[:, :, :2] | Y[:, :, :2] of outil.Homography(X, Y) (utils/outil.py:68-71) -- the input rows of rfx_host_dlt_null_vectors
 * (rfx_host_api.h).  off_host and xy_rows may be PINNED HOST memory (hipHostMalloc / torch pin_memory: device-visible, coherent):
 * the kernels then store straight into it and the host reads it after ONE event wait, no copy.  Rows >= row_cap are dropped
 * (the caller compares off[batch] with row_cap, grows the buffer and calls again).
 * rfx_ransac_patch_h_rows: the re-solved homographies, row off[b]+k of H_rows (9 f32 per row; device or pinned host memory),
 * into the hypotheses idx[b][k]; det gate re-evaluated.  Then stage 2. */
int rfx_ransac_degenerate_gather(const void* ws, const float* match1, const float* match2, const int32_t* n, int cap,
                                 const int64_t* samples, int N, int batch, int32_t* idx, int32_t* count, int32_t* off,
                                 int32_t* off_host, float* xy_rows, int row_cap, void* stream);
int rfx_ransac_patch_h_rows(void* ws, int cap, int N, int batch, const int32_t* idx, const int32_t* count, const int32_t* off,
                            const float* H_rows, int row_cap, void* stream);
/* rfx_dlt4_homography + the per-system flag byte (bit0 set; bit1: det(H) > 1e-6; bit2 (4): rank deficient). */
int rfx_dlt4_homography_flags(const float* X, const float* Y, int N, float* Hout, uint8_t* flags, void* stream);

/* Match lists of a batch (quick_start/coarseAlignFeatMatch.py:150-155): match1[b,i] = (xa[idx1[b,i]], ya[idx1[b,i]], 1),
 * match2[b,i] = (xb[idx2[b,i]], yb[idx2[b,i]], 1) for i < n[b], zeros after.  idx1/idx2 (batch,cap) int64 as written by
 * rfx_mutual_nn_batched_f32; xa/ya = source cell coordinates (getWHTensor "H"/"W" of all scales, utils/outil.py:21-24),
 * xb/yb = target cell coordinates; match1/match2 (batch,cap,3). */
int rfx_gather_matches_f32(const int64_t* idx1, const int64_t* idx2, const int32_t* n, int cap, const float* xa,
                           const float* ya, const float* xb, const float* yb, float* match1, float* match2, int batch,
                           void* stream);

/* ------------------------------------------------------------------------------------------
 * The rounds of the multi-homography drivers without host glue (evaluation/evalHpatch/evaluation.py:211-243,
 * evaluation/evalKITTI/evaluation.py:270-336).  All per-pair state (explained-region mask, homography count, cached match
 * list, result record) stays on the device; `active` (int32, n_active entries, NULL = identity) names the pairs of the batch
 * that still iterate, and every per-round array is indexed by the position k in that list.
 * ------------------------------------------------------------------------------------------ */
/* The RANSAC index draw on the device (utils/outil.py:120 draws torch.randint(0, nbMatch, (nbIter, nbPoint),
 * device=match1.device): on a GPU run the reference draws on the GPU).  samples (batch,N,4) int64;
 * samples[b,h,p] = word p of Philox4x32-10(counter = (h, id(b), stream_id lo, stream_id hi), key = (seed lo, seed hi))
 * modulo n[b] (torch's mapping of 32 random bits to a range below 2^32); n (batch) int32 ON THE DEVICE, n[b] <= 0 -> zeros.
 * id(b) = pair_ids[b] (batch int32 ON THE DEVICE: the caller's ABSOLUTE pair ids) or b when pair_ids is NULL.  With the
 * drivers' stream_id = (epoch << 32) | round, a pair's draws depend on (seed, its id, the round) only -- not on which other
 * pairs share the batch, are still active, or on how the stream is sharded over ranks.
 * No host value of nbMatch is needed: the draw is enqueued behind the kernel that produces the counts. */
int rfx_draw_samples_i64(const int32_t* n, int64_t* samples, int N, int batch, uint64_t seed, uint64_t stream_id,
                         const int32_t* pair_ids, void* stream);

/* (ABI 8) The keep map CoarseAlign.getCoarse of variants A / C multiplies the target features with before EVERY mutual matching
 * (quick_start/coarseAlignFeatMatch.py:136-143, evaluation/evalYFCC/coarseAlignFeatMatch.py:158-166), for the active pairs of a
 * batch: keep[k][cell] = 1.0 where (1 - fg) bilinear-resized (align_corners=False) to the (rt, ct) target feature map exceeds
 * 0.5, else 0.0; fg = ((mask + (1 - bg)) > 0.5) as in evaluation/evalYFCC/evaluation.py:239.  keep (n_active, rt*ct) is what
 * rfx_mutual_nn_batched_f32 takes as maskB -- the per-round re-matching of the YFCC driver shape stays on the device.
 * mask (batch,h,w) 0/1 floats or NULL together with bg == NULL (nothing explained, no background: all ones);
 * bg (batch,h,w) or NULL (= all ones); active (n_active) int32 or NULL (= 0..n_active-1). */
int rfx_keep_mask_f32(const float* mask, const float* bg, const int32_t* active, int n_active, int h, int w, int rt, int ct,
                      float* keep, void* stream);

/* CoarseAlign.getCoarse of the evaluation variant up to the match lists (evaluation/evalHpatch/coarseAlignFeatMatch.py:
 * 156-170) for the active pairs of a batch: fg = ((mask + (1 - bg)) > 0.5), MtExtend = 1 - fg bilinear-resized
 * (align_corners=False) to the (rt, ct) target feature map and thresholded at 0.5 -- evaluated only at the cells the cached
 * matches point to -- and the surviving matches compacted IN ORDER into match1/match2 (n_active,cap,3) as
 * rfx_gather_matches_f32 lays them out (rows >= n_out[k] zero); n_out (n_active) int32; kept (n_active,cap) int32 or NULL =
 * slot in the cached list of every surviving match (-1 padding).  idx1/idx2 (batch,cap) int64 + count (batch) int32 = the
 * cached mutual matches (rfx_mutual_nn_batched_f32); mask (batch,h,w) 0/1 floats; bg (batch,h,w) or NULL (= all ones). */
int rfx_filter_matches_f32(const int64_t* idx1, const int64_t* idx2, const int32_t* count, int cap, const int32_t* active,
                           int n_active, const float* mask, const float* bg, int h, int w, int rt, int ct, const float* xa,
                           const float* ya, const float* xb, const float* yb, float* match1, float* match2, int32_t* n_out,
                           int32_t* kept, void* stream);

/* The accept rule, mask update and result-record store of one round.  match (n_active,h,w) = PredFlowMask's matchability
 * (after the small-component filter for KITTI); mask (batch,h,w) in/out; ransac_result (n_active,4) from
 * rfx_ransac_h4_batched; n_match (n_active); nbH (batch) int32 in/out = homographies accepted so far.
 *   mode 0 (evalHpatch/evaluation.py:225,238-239): gain = mean(match * (1 - fg)); mask <- (mask + match * (1 - fg)) >= 1
 *   mode 1 (evalKITTI/evaluation.py:322,332-333):  gain = mean((match > 0.9999) * (1 - fg)); mask <- (... ) > 0.9999
 * accept[k] = n_match >= 4 && status == 0 && (gain > th || nbH == 0); gain is the float64 sum of the float32 terms divided by
 * h*w and rounded to float32 (numpy's float32 mean up to summation order).  For an accepted pair the record row
 * rec + b*rec_stride (floats; NULL = no records) receives, at slot s = nbH[b] < max_h: H -> [off_H + 9 s], flowDown8 (2,h8,w8)
 * -> [off_flow + 2 h8 w8 s], match12Down8 | match21Down8 -> [off_match + 2 h8 w8 s], flowD2 (2,hd2,wd2; NULL outside KITTI) ->
 * [off_d2 + 2 hd2 wd2 s], and element 0 = the new homography count; then nbH[b] += 1.  The host reads back `accept`
 * (n_active int32) -- the ONE device-to-host copy of a round.  ws: rfx_multih_accept_ws_bytes(n_active). */
size_t rfx_multih_accept_ws_bytes(int n_active);
int rfx_multih_accept_f32(const float* match, float* mask, const float* bg, const int32_t* active, int n_active, int h, int w,
                          const int32_t* ransac_result, const int32_t* n_match, int32_t* nbH, double th, int mode,
                          int32_t* accept, float* gain, void* ws, const float* bestH, const float* flowDown8,
                          const float* match12Down8, const float* match21Down8, int h8, int w8, const float* flowD2, int hd2,
                          int wd2, float* rec, long long rec_stride, int max_h, int off_H, int off_flow, int off_match,
                          int off_d2, void* stream);

/* ------------------------------------------------------------------------------------------
 * rfx_conv1x1_split_f32 (ABI 10): the 1x1 / stride 1 / pad 0 convolution of rfx_conv2d_f32 (the Bottleneck conv1 / conv3 layers,
 * model/resnet50.py:71-79,93-103) with float32 results computed on the bf16 matrix cores by EXACT operand splitting:
 *     x = hi + mid + lo,  hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)      (round to nearest even; the sum is exact)
 *     out[n,m,p] = act(scale[m] * S + shift[m] + residual),
 *     S = sum_k (wh xh) + sum_k (wh xm + wm xh + wm xm + wh xl + wl xh)                 (every product exact in float32; the two
 *         sums in two float32 accumulators that round once per 16 k, added once at the end; the three dropped terms are below
 *         2^-32 |w x|).  Error against the float64 sum: 0.4x the fp32 kernel's fma chain (profiles/r06_bf16_split_study.json).
 * NOT bit-identical to rfx_conv2d_f32 (closer to the exact sum); +-inf inputs give NaN.  Cin % 16 == 0.
 * wS: the weights split on the host and packed in fragment order, bf16 bit patterns (uint16):
 *     wS[kb = k / 16][piece (hi, mid, lo)][h = (k % 16) / 8][m (Mpad = Cout rounded up to 128, rows >= Cout zero)][k % 8]
 * in (N,Cin,HW), out / residual (N,Cout,HW) float32.
 * ------------------------------------------------------------------------------------------ */
int rfx_conv1x1_split_f32(const float* in, const void* wS, const float* scale, const float* shift, const float* residual,
                          float* out, int N, int Cin, int HW, int Cout, int act, void* stream);
/* ... with a stride (1 or 2; pad 0): the projection shortcuts of the trunk (model/resnet50.py:139-143): out (N,Cout,Ho,Wo),
 * Ho = (Hin - 1) / stride + 1, reads input pixel (y * stride, x * stride). */
int rfx_conv1x1_split_strided_f32(const float* in, const void* wS, const float* scale, const float* shift, const float* residual,
                                  float* out, int N, int Cin, int Hin, int Win, int Cout, int stride, int act, void* stream);
/* rfx_conv3x3_split_f32 (ABI 10): the same scheme for the 3x3 / stride 1 / pad 1 convolution (ResNet-50 layer3 conv2,
 * model/resnet50.py:75; the FeatureExtractor's BasicBlock convolutions, model/model.py:32-35; conv2 / conv3 of the NetFlowCoarse /
 * NetMatchability stacks, model/model.py:170-181): nine shifted 1x1 products over one staged, split halo patch.  Cin % 16 == 0.
 * wS3: bf16 bit patterns, [kb = c / 16][tap = kh * 3 + kw][piece][h = (c % 16) / 8][m (Mpad)][c % 8] = piece of W[m][c][kh][kw].
 * in (N,Cin,H,W), out / residual (N,Cout,H,W) float32.  NOT bit-identical to rfx_conv3x3_f32 / rfx_conv2d_f32 (closer to the exact sum). */
int rfx_conv3x3_split_f32(const float* in, const void* wS3, const float* scale, const float* shift, const float* residual,
                          float* out, int N, int Cin, int H, int W, int Cout, int act, void* stream);
/* ... stride 2 (pad 1): ResNet-50 layer2.0 / layer3.0 conv2 (model/resnet50.py:75), the FeatureExtractor's strided conv1
 * (model/model.py:32).  out (N,Cout,Ho,Wo), Ho = (H - 1) / 2 + 1; same wS3; Cin % 16 == 0; 128-channel tiles. */
int rfx_conv3x3_split_s2_f32(const float* in, const void* wS3, const float* scale, const float* shift, const float* residual,
                             float* out, int N, int Cin, int H, int W, int Cout, int act, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sky segmentation forward pass: SegNet.getSky (segNet/segEval.py:23-43) = ResNet-50-dilated encoder
 * (segNet/segModel.py:156-215: layer3 / layer4 with dilation 2 / 4 instead of stride) + PPM decoder (:218-265) over five
 * scales, averaged, arg-max, (pred == segId).  Called once per target image by CoarseAlign.skyFromSeg
 * (evaluation/evalHpatch/coarseAlignFeatMatch.py:63-64,152-153; evaluation/evalHpatch/evaluation.py:177-180).
 * The convolutions run on the convolution family above plus ONE more geometry:
 *
 * rfx_conv2d_dilated_f32: rfx_conv2d_f32 with `dilation` > 1 (segModel.py:196-205: 3x3, dilation = padding = 2 or 4):
 *     out[n,m,oh,ow] = act(scale[m] * sum in[n,c,oh*s-p+kh*d,ow*s-p+kw*d] * w[m,c,kh,kw] + shift[m] + residual)
 * wT as for rfx_conv2d_f32; ktab holds the DILATED offsets: (c<<8)|((kh*d)<<4)|(kw*d), (KH-1)*d + 1 <= 15.  Same kernels, same
 * summation order as rfx_conv2d_f32 (the gather adds the table's offsets to the window origin as they are).
 * ------------------------------------------------------------------------------------------ */
int rfx_conv2d_dilated_f32(const float* in, const float* wT, const int32_t* ktab, const float* scale,
                           const float* shift, const float* residual, float* out, int N, int Cin, int Hin,
                           int Win, int Cout, int KH, int KW, int stride, int pad, int dilation, int act, void* stream);
/* nn.AdaptiveAvgPool2d((Hout, Wout)) on NC planes (segModel.py:227): window [floor(o*In/Out), ceil((o+1)*In/Out)), row-major sum / count. */
int rfx_adaptive_avgpool2d_f32(const float* in, float* out, int NC, int Hin, int Win, int Hout, int Wout, void* stream);
/* scores[n,c,p] = (accumulate ? scores[n,c,p] : 0) + softmax_c(logits[n,:,p]) / div      (segModel.py:258 + segEval.py:34-35:
 * ``scores = scores + pred_tmp / 5``); logits / scores (N,C,HW). */
int rfx_softmax_accum_f32(const float* logits, float* scores, int N, int C, long long HW, float div, int accumulate, void* stream);
/* torch.max(scores, dim=1) -> mask[n,p] = (pred == id) (complement != 0: 1 - that), float32 (segEval.py:37-43); pred (N,HW) int32
 * optional (NULL: not stored).  First maximum wins. */
int rfx_argmax_mask_f32(const float* scores, int N, int C, long long HW, int id, int complement, float* mask, int32_t* pred,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RFX_API_H */
