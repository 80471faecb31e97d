/* szn.h -- C-ABI of libszn_hip.so: the MI355X (gfx950) kernels underneath the SZN pixel-embedding
 * training path.
 *
 * The reference (RohanDoshi2018/ZeroshotSemanticSegmentation) has no FFI of its own: its "operator
 * interface" for this path is the set of torch calls made by models.py / utils.py / train.py.  Each
 * entry point below names the reference call site (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Every pointer is a CALLER-OWNED DEVICE pointer
 *     (the library never allocates, frees or retains them) unless the comment says "host".
 *   - every function returns int: 0 = SZN_OK, <0 = error; szn_last_error() gives the text
 *     (thread-local).  No host synchronisation happens inside any entry point; every launch goes
 *     to the hipStream_t passed as `stream` (pass torch.cuda.current_stream().cuda_stream).
 *   - activations are NHWC [B][H][W][C] ("pixel-major"), conv weights OHWI [Cout][KH][KW][Cin]
 *     (== torch channels_last storage of an (O,I,KH,KW) tensor), element type given by `dtype`
 *     (SZN_F32 parity path / SZN_BF16 throughput path / SZN_F16 IEEE-half variant); accumulation is always fp32.
 *   - the network-boundary tensors keep the reference's layout: input image (B,3,H,W) NCHW f32,
 *     score (B,E,H,W) NCHW f32, labels (B,H,W) int64 with -1 = ignore.
 */
#ifndef SZN_H_
#define SZN_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* szn_stream_t; /* hipStream_t */

enum { SZN_OK = 0, SZN_ERR_ARG = -1, SZN_ERR_LAUNCH = -2, SZN_ERR_UNSUPPORTED = -3 };
enum { SZN_F32 = 0, SZN_BF16 = 1, SZN_F16 = 2 };   /* SZN_F16: IEEE half activations / weight images (BASELINE configs[4]) */

/* a set of class indices below SZN_MAX_CLASSES: bit (k % 64) of w[k / 64] = class k.  Passed by HOST pointer, read during the
 * call (NULL = the empty set); used by the *_k entry points (more than 64 classes).                                                                                        */
#define SZN_MAX_CLASSES 256
typedef struct szn_class_set {
    uint64_t w[4];
} szn_class_set;

/* ---- library -------------------------------------------------------------------------------- */
const char* szn_last_error(void);
/* Name of the kernel most recently launched by this thread's last call (e.g. "conv3x3_regw", "conv_igemm_wide",
 * "wgrad_taps_reduce"): which specialised path the dispatcher took.  Test / profiling aid -- the parity tests assert it
 * so that a specialised kernel cannot silently fall back to the generic one. */
const char* szn_last_kernel(void);
const char* szn_prev_kernel(void);   /* the launch before it (e.g. the GEMM kernel in front of a split-K epilogue) */
int szn_version(void); /* major*10000 + minor*100 + patch */
/* The environment knobs this build reads (A/B switches and dispatch thresholds; defaults = the shipped behaviour): szn_knob_count() names,
 * szn_knob_name(i) for 0 <= i < count (NULL outside).  The library refuses to read a variable that is not in this list, and
 * tests/test_gpu_knobs.py runs a training step under a non-default value of every one of them.                                              */
int szn_knob_count(void);
const char* szn_knob_name(int i);
typedef struct {
    char name[128];
    char arch[32];
    int compute_units;
    int wavefront;
    int lds_bytes_per_block;
    int64_t hbm_bytes;
    int clock_mhz;
} szn_device_info_t;
int szn_device_info(int device, szn_device_info_t* out /* host */);
/* A stream confined to the compute units of `mask` (bit i of word i / 32 = CU i; n_words x 32 bits) -- what
 * torch.cuda.Stream() would be for the reference if it split its one-image step (train.py:82-84) over two queues: the engine
 * runs fc6's HBM-bound weight gradient + Adam step there while the few-tile dgrads of conv5_x .. conv3_x use the other CUs.
 * The handle is a hipStream_t: wrap it (torch.cuda.ExternalStream) or pass it as `stream` to any entry point. */
int szn_stream_create_cu_mask(int n_words, const uint32_t* mask /* host */, szn_stream_t* out /* host */);
int szn_stream_destroy(szn_stream_t stream);

/* ---- stride-1 convolution as implicit GEMM on MFMA --------------------------------------------
 * Replaces nn.Conv2d forward/backward for conv1_2..conv5_3 (3x3 pad 1), fc6 (7x7 valid), fc7 and
 * score_fr || seenmask_score (1x1): models.py:45-97 (construction), models.py:117-149 (forward),
 * and their autograd backward triggered by trainer_fcn.py:157 / trainer_seenmask.py:81.
 * Ci must be a multiple of 64 (bf16) / 32 (f32); Co is arbitrary.                               */
/* What a call decided that its caller has to know afterwards (HOST memory, optional): filled before the entry point returns.
 * colsum_rows   = the partial rows the call wrote into its colsum slab (0: it used no slab) -- a function of the kernel the dispatcher picked;
 *                 szn_colsum_reduce_batch takes it.
 * work_fraction = 1.0, or the fraction of the dense tiles the call executed under the constant-border hint (cb_on).                        */
typedef struct szn_call_result {
    int colsum_rows;
    float work_fraction;
} szn_call_result_t;

typedef struct {
    int dtype;        /* SZN_F32 | SZN_BF16: element type of in / w / gate / out                 */
    int B, Hi, Wi, Ci; /* input  [B][Hi][Wi][Ci], pixel stride ldi >= Ci elements                 */
    int Ho, Wo, Co;   /* output [B][Ho][Wo][Co], pixel stride ldo >= Co; Ho = Hi + 2*pad - KH + 1 */
    int KH, KW, pad;
    int ldi, ldo, ldg; /* pixel strides (elements) of in / out / gate                            */
    int relu;         /* epilogue max(v,0)  (nn.ReLU, models.py:44..90)                           */
    int out_f32;      /* store out as float even when dtype == SZN_BF16                           */
    void* workspace;  /* optional device scratch for split-K (few output tiles, long K: fc6/fc7); */
    size_t workspace_bytes; /* used when >= B*Ho*Wo*Co*4 bytes (dgrad: B*Hi*Wi*Ci*4); NULL = never split */
    float* colsum;    /* optional f32 [Co] (dgrad: [Ci]): += column sums of the tensor being written, i.e. the
                         bias gradient of the layer that produced the gated input (saves a pass over it)  */
    void* pool_out;   /* szn_conv2d_fwd only, optional: ALSO write MaxPool2d(2, 2, ceil_mode=True) of the output
                         (models.py:47,54,63,72,81 follow a ReLU'd conv): [B][ceil(Ho/2)][ceil(Wo/2)][Co], same element
                         type as out, dense (needs ldo == Co, relu != 0).  Fused into the conv epilogue where the
                         kernel supports it (the 710^2 / 355^2 layers), else szn_maxpool2x2_ceil_fwd runs behind it */
    float* colsum_slab; /* optional fp32 workspace [colsum_slab_rows][Co] (dgrad: [..][Ci]), 16-B aligned, declared at the END of
                         the struct.  With it the kernel does NOT touch colsum: every pixel tile / persistent block writes its
                         partial column sums into its own row (result->colsum_rows rows, a function of the kernel the
                         dispatcher picked) and the caller adds them with szn_colsum_reduce_batch in a fixed order --
                         bit-reproducible bias gradients.  Without it the partials are added to colsum with fp32 atomics
                         (same value up to the order of the additions).                                            */
    int colsum_slab_rows; /* rows the slab can hold; an error is returned if the kernel needs more                  */
    void* pool_code;    /* optional, with pool_out: one byte per pooled element [B][ceil(Ho/2)][ceil(Wo/2)][Co] = position 2 dy + dx
                         of the FIRST maximum of its window (torch's scan order), or 4 when that maximum is not positive (the
                         ReLU gate): everything szn_maxpool2x2_ceil_bwd_code needs.  8-B aligned.                            */
    int pool_only;      /* with pool_out: the caller does not need the un-pooled tensor (in training it is read once, by the pool's
                         backward pass, which takes pool_code instead: 516 + 258 MB per step that are neither written nor read back);
                         `out` must still be valid memory, a kernel that fuses the pool MAY leave it unwritten             */
    int cb_on;          /* constant-border hint (optional; a kernel that does not take it computes everything).  models.py:43 pads conv1_1
                         by 100, so on the 710^2 / 355^2 maps most pixels outside the image's reach hold ONE value per channel (conv1_1
                         writes relu(bias) there) and stay that way through conv1_2 .. conv2_2 away from the tensor edge.
                         szn_conv2d_fwd: cb_rect = output rows [r0, r1) x columns [c0, c1) the image can influence; cb_const = output
                         rows / columns [r0, r1) x [c0, c1) outside of which the zero padding of the layers so far is felt.  The caller
                         asserts that the INPUT is constant accordingly, so output pixels inside cb_const and outside cb_rect are all
                         equal: a kernel that takes the hint (conv3x3_regw) runs its tiles over cb_rect and the edge frame only, and
                         broadcasts one computed pixel to the rest (out / pool_out / pool_code alike).  Same bits as the dense
                         computation.  result->work_fraction = the fraction of the dense tiles the call executed.
                         szn_conv2d_dgrad with a gate (coordinates of DIN): cb_rect = the rows x columns outside of which every pixel of
                         the gate tensor equals gate pixel (row r0 - 1, column c0) of image 0 (needs r0 >= 1) -- the kernel may read that
                         pixel instead; cb_const = the rows x columns of DIN the caller is going to read -- stores outside may be
                         skipped (DIN is undefined there; colsum still covers every pixel).  conv1_2's dgrad: its output feeds only
                         szn_conv1_1_wgrad, whose read set szn_conv1_1_wgrad_reads() reports.
                         szn_conv2d_wgrad (coordinates of the INPUT x): cb_rect = the rows x columns the image can influence, cb_const =
                         the rows x columns outside of which zero padding is felt; the caller asserts that every pixel of x inside
                         cb_const and outside cb_rect holds the same value per channel.  conv_wgrad_taps then runs only the 16 x 16
                         output tiles whose input patch is not constant; the others contribute (their column sum of dout) x (that
                         pixel) to all nine taps -- the same sum in a different order (fp32; <= 1e-5 relative against the dense
                         result, tests/test_gpu_conv.py), ignored with accumulate != 0.  result->work_fraction reports it.
                         `colsum` (otherwise unused by szn_conv2d_wgrad) may then hold that column sum [Co], computed by the producer
                         of dout over the region szn_conv2d_wgrad_cb_region() names (szn_maxpool2x2_ceil_bwd_code_cb); NULL = the
                         call sums them itself.                                                                                   */
    int cb_rect[4];
    int cb_const[4];
    int reserved_cus;   /* optional (0 = none): compute units the persistent one-block-per-CU kernels (conv3x3_regw, conv_wgrad_taps)
                         leave idle for another queue -- under data parallelism the RCCL all-reduce of the gradient buckets runs on
                         its own stream while dgrad / wgrad continue (engine.GradBuckets; train.py has no counterpart: the reference
                         is single-GPU).  Tiled kernels ignore it (their blocks come and go; the hardware interleaves the queues).  */
    void* dw_lp;        /* szn_conv2d_wgrad only, optional: deliver the weight gradient as a 16-bit image (dw_lp_dtype = SZN_BF16 |
                         SZN_F16, OHWI like dw, 8-B aligned) -- the wire format of the data-parallel gradient exchange
                         (engine.GradBuckets, SZN_GRAD_COMM=bf16; the reference is single-GPU: trainer_fcn.py:157-158 call
                         loss.backward(); optim.step() back to back).  The kernels that own the final store of a gradient element
                         (wgrad_taps_reduce, conv_wgrad_wide's epilogue, wgrad_slab_reduce) round it once from the fp32 sum and write
                         2 B instead of 4; dw must still be valid fp32 memory of the full size (scratch for the paths that finish in
                         fp32 and convert), its content is UNDEFINED afterwards.  Ignored with accumulate != 0 (error).          */
    int dw_lp_dtype;
    szn_call_result_t* result; /* optional HOST pointer: szn_conv2d_fwd / _dgrad / _wgrad / _wgrad_adam report through it (no "last call" state) */
} szn_conv_desc_t;

/* out[m][n] = epi( sum_k in(m,k) * w[n][k] + bias[n] )
 * epi(v):  relu -> (gate ? (gate[m][n] > 0 ? v : 0) : v) -> (chan_scale ? v*chan_scale[b][n] : v)
 * bias, gate, chan_scale may be NULL.  chan_scale is the Dropout2d factor per (image, channel)
 * (models.py:86,91: 0 or 1/(1-p)).                                                               */
int szn_conv2d_fwd(const szn_conv_desc_t* d, const void* in, const void* w, const float* bias,
                   const void* gate, const float* chan_scale, void* out, szn_stream_t stream);

/* wT[ci][KH-1-kh][KW-1-kw][co] = w[co][kh][kw][ci]: the weight image szn_conv2d_dgrad consumes.  */
int szn_pack_weight_dgrad(int dtype, int Co, int KH, int KW, int Ci, const void* w, void* wT,
                          szn_stream_t stream);
/* the same for n layers in ONE launch (the per-step refresh after the optimizer, train.py:126-133: most layers are a few
 * tiles and would each pay a dispatch): 16-bit images, square K x K filters, Co and Ci multiples of 64, n <= 24.          */
int szn_pack_weight_dgrad_batch(int dtype, int n, const void* const* w, void* const* wT, const int* Co, const int* K,
                                const int* Ci, szn_stream_t stream);

/* d (forward geometry) -> din[B][Hi][Wi][Ci] = conv(dout, wT) with pad' = KH-1-pad; epilogue as
 * above with gate = the forward INPUT activation (ReLU backward) and chan_scale = the dropout
 * factors that were applied to that input.  d->ldo is the pixel stride of DOUT, d->ldi of DIN.   */
int szn_conv2d_dgrad(const szn_conv_desc_t* d, const void* dout, const void* wT, const void* gate,
                     const float* chan_scale, void* din, szn_stream_t stream);
/* cb_on == 2 on a gated szn_conv2d_dgrad with colsum (conv1_2's dgrad: din feeds only szn_conv1_1_wgrad and conv1_1's bias gradient): the
 * tiles that neither store anything (outside cb_const) nor see a varying gate (outside cb_rect) are not run; colsum then covers the
 * tiles that ran, and szn_conv2d_dgrad_border_finish adds the rest from region sums of dout -- it is linear in dout there (one gate value
 * per channel), so it needs skip_sum [Co] = sum of dout over the map outside szn_conv2d_dgrad_border_region()'s inner rectangle (from the
 * producer of dout: szn_maxpool2x2_ceil_bwd_code_cb) and 1-pixel strips it reads itself.  Same value up to fp32 summation order.
 * The caller checks result->work_fraction < 1 after the dgrad call before it calls finish (a kernel that ignores the hint computes the
 * complete column sums).  workspace: 2 * 24 * B * Co floats, 16-B aligned.                                                                */
int szn_conv2d_dgrad_border_region(const szn_conv_desc_t* d, int region[8]);
int szn_conv2d_dgrad_border_finish(const szn_conv_desc_t* d, const void* dout, const void* wT, const void* gate,
                                   const float* skip_sum, float* colsum, void* workspace, szn_stream_t stream);

/* dgrad of a large-window convolution (models.py:84 fc6 = Conv2d(512, 4096, 7) backward) as GEMM + col2im: as a
 * convolution over the padded dout map szn_conv2d_dgrad executes 1.83x the algorithmic FLOPs of fc6's dgrad (most taps
 * of the border pixels are padding); here
 *   Y[(b,oh,ow)][(kh,kw,ci)] = sum_co dout[(b,oh,ow)][co] * w[co][kh][kw][ci]   (fp32, in d->workspace)
 *   din[b][ih][iw][ci]       = sum_{kh,kw} Y[(b, ih+pad-kh, iw+pad-kw)][(kh,kw,ci)]
 * wG = the plain transpose [KH*KW*Ci][Co] of the OHWI filter bank, i.e. szn_pack_weight_dgrad(dtype, Co, 1, 1,
 * KH*KW*Ci, w, wG).  d = forward geometry (d->ldo = pixel stride of dout, d->ldi of din); d->workspace must hold
 * szn_conv2d_dgrad_gemm_workspace_bytes(d) = B*Ho*Wo*KH*KW*Ci*4 bytes.  Co must be a multiple of 64 (bf16) / 32
 * (f32) like every reduction dimension of szn_conv2d_*; Ci a multiple of 4.  No gate / chan_scale / colsum epilogue
 * (fc6's input is a pooled map: the ReLU gate is applied by szn_maxpool2x2_ceil_bwd).                          */
size_t szn_conv2d_dgrad_gemm_workspace_bytes(const szn_conv_desc_t* d);
int szn_conv2d_dgrad_gemm(const szn_conv_desc_t* d, const void* dout, const void* wG, void* din,
                          szn_stream_t stream);
/* The same dgrad on the filter bank in its FORWARD layout w [Co][KH][KW][Ci] (no transposed copy per optimizer step: for fc6 that copy
 * is 205 MB in + 205 MB out): Y = dout x w runs on the two-K-major-operand kernel of the wide weight gradient with A = dout^T (only the
 * small dout is transposed, into the workspace).  16-bit dtypes, dense dout (ldo == Co), B*Ho*Wo a multiple of 8 and >= 256, enough
 * 256 x 256 tiles: szn_conv2d_dgrad_gemm_native_supported(d) says whether this shape qualifies (callers fall back to
 * szn_conv2d_dgrad_gemm otherwise).  Workspace: szn_conv2d_dgrad_gemm_native_workspace_bytes(d) (Y fp32 | dout^T).             */
int szn_conv2d_dgrad_gemm_native_supported(const szn_conv_desc_t* d);
size_t szn_conv2d_dgrad_gemm_native_workspace_bytes(const szn_conv_desc_t* d);
int szn_conv2d_dgrad_gemm_native(const szn_conv_desc_t* d, const void* dout, const void* w, void* din,
                                 szn_stream_t stream);

/* dw[co][kh][kw][ci] (+)= sum_pixels dout[p][co] * in[p shifted by (kh,kw)][ci]   (fp32, OHWI)
 * accumulate != 0 adds into dw (split-K partial sums are added with fp32 atomics; dw must then be
 * zero or hold a running gradient); accumulate == 0 zeroes dw first on the same stream.           */
int szn_conv2d_wgrad(const szn_conv_desc_t* d, const void* in, const void* dout, float* dw,
                     int accumulate, szn_stream_t stream);
/* Weight gradient + Adam step of the layer's weights in ONE launch (fc6 / fc7, i.e. the layers szn_conv2d_wgrad gives to
 * conv_wgrad_wide: szn_conv2d_wgrad_adam_supported == 1).  Equivalent to szn_conv2d_wgrad(accumulate = 0) followed by szn_adam_step
 * over the layer's slice (train.py:130-133,174-175: loss.backward(); optim.step()), bit for bit; only valid where the gradient
 * is final when the kernel ends (one rank, no gradient accumulation, no dynamic loss scale).  param / exp_avg / exp_avg_sq (f32)
 * and w_lp (optional image of the compute dtype) in the gradient's OHWI order; dw may be NULL (gradient not stored).
 * SZN_ERR_UNSUPPORTED (nothing launched) for other layers.                                                              */
typedef struct szn_adam_args {
    float* param; float* exp_avg; float* exp_avg_sq;
    void* w_lp; int w_lp_dtype;
    float lr, beta1, beta2, eps, weight_decay;
    int step;              /* 1-based, like szn_adam_step */
    float grad_scale;
    int grad_optional;     /* 1: the caller does not read dw afterwards: the gradient is not stored even when dw != NULL */
} szn_adam_args_t;
int szn_conv2d_wgrad_adam_supported(const szn_conv_desc_t* d);
int szn_conv2d_wgrad_adam(const szn_conv_desc_t* d, const void* x, const void* dout, float* dw,
                          const szn_adam_args_t* opt, szn_stream_t stream);

/* db[n] (+)= sum_m dout[m][n], m < M rows with pixel stride ldd.                                  */
int szn_bias_grad(int dtype, long M, int Co, int ldd, const void* dout, float* db, int accumulate,
                  szn_stream_t stream);
/* the same with a slab for the per-block partial rows ([colsum_slab_rows][Co] fp32, see szn_conv_desc_t.colsum_slab): db is
 * only zeroed (accumulate == 0), the sums arrive through szn_colsum_reduce_batch.  colsum_slab == NULL: as szn_bias_grad. */
int szn_bias_grad_slab(int dtype, long M, int Co, int ldd, const void* dout, float* db, int accumulate,
                       float* colsum_slab, int colsum_slab_rows, int* colsum_rows_out /* host, optional: rows written */, szn_stream_t stream);
/* Deterministic bias gradients.  The kernels that produce column sums (szn_conv2d_fwd / _dgrad with colsum,
 * szn_maxpool2x2_ceil_bwd, szn_bias_grad_slab) write partial rows into the caller's slab when one is given;
 * every such call reports the number of rows it wrote (0: it used no slab) through its own out-parameter (szn_conv_desc_t.result /
 * colsum_rows_out), and szn_colsum_reduce_batch adds out[j][c] += sum_{r < rows[j]} slabs[j][r * C[j] + c] for n jobs in one launch, rows in
 * ascending order (host arrays of length n; every bias gradient of a backward pass in ONE launch).                  */
int szn_colsum_reduce_batch(int n, const float* const* slabs, const int* rows, const int* C, float* const* out,
                            szn_stream_t stream);

/* the "pixel projection" of the north star: score_fr (|| seenmask_score) 1x1 conv, models.py:93,97,
 * 145,149.  Thin aliases of the conv entry points with KH=KW=1, pad=0 (M = B*h*w rows).           */
int szn_gemm_proj_fwd(int dtype, long M, int K, int N, int ldo, const void* x, const void* w,
                      const float* bias, float* out_f32, szn_stream_t stream);
int szn_gemm_proj_dgrad(int dtype, long M, int K, int N, int ldd, const void* dout, const void* wT,
                        const void* gate, const float* chan_scale, void* dx, szn_stream_t stream);
int szn_gemm_proj_wgrad(int dtype, long M, int K, int N, int ldd, const void* x, const void* dout,
                        float* dw, int accumulate, szn_stream_t stream);

/* ---- conv1_1: 3 -> 64 channels, 3x3, pad 100, reads the NCHW f32 image directly ----------------
 * models.py:43,116 (+ ReLU models.py:44).  w is OHWI f32 [64][3][3][3], out NHWC dtype.
 * dtype SZN_F32: exact fp32 products (v_mfma_f32_16x16x4_f32).  SZN_BF16 / SZN_F16: image values and filters are rounded to the
 * storage type before the 16-bit MFMA, fp32 accumulation -- the arithmetic of every other layer on those paths (max error against
 * fp32 on pixel-valued inputs 4e-3 / 6e-4 of the output range); SZN_CONV1_1_F32MMA=1 keeps fp32 operands there too.             */
int szn_conv1_1_fwd(int dtype, int B, int H, int W, int pad, const float* x_nchw, const float* w,
                    const float* bias, void* out, szn_stream_t stream);
/* The same forward pass writing a CROPPED map (round 5): cut = {ya, ye, ya2, ye2, xa, xe, xa2, xe2}, output rows [ya, ye) and [ya2, ye2) and
 * columns [xa, xe), [xa2, xe2) are not stored -- the constant band the caller removes in front of conv1_2 (models._band_cut; models.py:43 pads
 * by 100, so those rows hold relu(bias) and a 3x3 convolution behind them needs only a few of them); out is [B][Hc][Wc][64].  Staged 16-bit
 * kernel only (SZN_ERR_UNSUPPORTED otherwise: crop the full map with szn_band_remap).  szn_conv1_1_wgrad_c reads its dout in that
 * cropped layout (the removed pixels see no image pixel and contribute nothing); db is not produced (column sums of conv1_2's dgrad).   */
int szn_conv1_1_fwd_c(int dtype, int B, int H, int W, int pad, const float* x_nchw, const float* w,
                      const float* bias, void* out, const int cut[8], szn_stream_t stream);
int szn_conv1_1_wgrad_c(int dtype, int B, int H, int W, int pad, const float* x_nchw, const void* dout, float* dw,
                        int accumulate, void* workspace, const int cut[8], szn_stream_t stream);
/* dw[64][3][3][3], db[64] from dout (already ReLU-gated) -- no dgrad: the image needs no gradient.
 * bf16: a fused MFMA kernel (taps gathered from the image in registers, padding-only pixels skipped, fp32 slabs in
 * `workspace`, deterministic); f32: an MFMA wgrad over the im2col image (27 taps padded to 32) in `workspace`.  */
size_t szn_conv1_1_wgrad_workspace_bytes(int dtype, int B, int H, int W, int pad);
int szn_conv1_1_wgrad(int dtype, int B, int H, int W, int pad, const float* x_nchw,
                      const void* dout, float* dw, float* db, int accumulate, void* workspace,
                      szn_stream_t stream);
/* The part of dout a szn_conv1_1_wgrad call with db == NULL and these arguments reads: rect = rows [r0, r1) x columns [c0, c1) of
 * the (H + 2 pad - 2) x (W + 2 pad - 2) map.  Returns 1 when that is a proper sub-rectangle (the fused 16-bit kernel skips the pixels
 * whose windows hold padding only), 0 when the call reads everything (rect = the whole map).  What the producer of dout -- conv1_2's
 * dgrad -- may leave unwritten: szn_conv_desc_t.cb_const.                                                                          */
int szn_conv1_1_wgrad_reads(int dtype, int B, int H, int W, int pad, int rect[4]);

/* ---- MaxPool2d(2, stride 2, ceil_mode=True): models.py:47,54,63,72,81 ---------------------------- */
int szn_maxpool2x2_ceil_fwd(int dtype, int B, int Hi, int Wi, int C, const void* in, void* out,
                            szn_stream_t stream);
/* din = relu_bwd(pool_bwd(dout)): the gradient goes to the FIRST maximum of each window (torch
 * scan order) and is then gated by in > 0 (the ReLU that precedes every pool).                    */
int szn_maxpool2x2_ceil_bwd(int dtype, int B, int Hi, int Wi, int C, const void* in,
                            const void* out, const void* dout, void* din, float* colsum /* [C] += sum of din, or NULL */,
                            float* colsum_slab /* optional, see szn_conv_desc_t.colsum_slab */, int colsum_slab_rows,
                            int* colsum_rows_out /* host, optional: partial rows written into the slab */, szn_stream_t stream);
/* The pair that works from winner codes (one byte per pooled element: 0 .. 3 = position 2 dy + dx of the first maximum, 4 = maximum
 * not positive) instead of the pool's input: the forward pass writes them (here, or fused into the producing conv through
 * szn_conv_desc_t.pool_code), the backward pass reads d(pooled) + codes only.  Same gradients bit for bit.                     */
int szn_maxpool2x2_ceil_fwd_code(int dtype, int B, int Hi, int Wi, int C, const void* in, void* out, void* code /* or NULL */,
                                 szn_stream_t stream);
int szn_maxpool2x2_ceil_bwd_code(int dtype, int B, int Hi, int Wi, int C, const void* code, const void* dout, void* din,
                                 float* colsum, float* colsum_slab, int colsum_slab_rows, int* colsum_rows_out, szn_stream_t stream);
/* The same pass with dout given in the coordinates of the NEXT conv block's cropped input, [B][Hs][Ws][C] (szn_band_remap): the transposed band map is
 * applied while reading -- pooled pixel (oh, ow) takes the fp32 sum of source rows ytab[oh] = {start, count} x columns xtab[ow] = {start, count}
 * (device int tables of (Hi + 1) / 2 and (Wi + 1) / 2 pairs; almost every pair is {shifted index, 1}), rounded once.  Replaces the two szn_band_remap
 * passes the engine ran in front of szn_maxpool2x2_ceil_bwd_code at every cropped block boundary. */
int szn_maxpool2x2_ceil_bwd_code_gather(int dtype, int B, int Hi, int Wi, int C, const void* code, const void* dsrc, int Hs, int Ws,
                                        const int* ytab, const int* xtab, void* din, float* colsum, float* colsum_slab,
                                        int colsum_slab_rows, int* colsum_rows_out, szn_stream_t stream);
/* The same, also summing din over the regions its consumers -- the conv in front of the pool's weight gradient and dgrad -- do not run
 * tile by tile under the constant-border hint: skip_regions [n_regions][8] (pixels, even) from szn_conv2d_wgrad_cb_region() /
 * szn_conv2d_dgrad_border_region(), n_regions = 1 or 2, skip_sum [n_regions][C] out (szn_conv_desc_t.colsum of the weight-gradient call /
 * skip_sum of szn_conv2d_dgrad_border_finish), skip_slab [n_regions][colsum_slab_rows][C] scratch.  colsum / colsum_slab required. */
int szn_maxpool2x2_ceil_bwd_code_cb(int dtype, int B, int Hi, int Wi, int C, const void* code, const void* dout, void* din,
                                    float* colsum, float* colsum_slab, int colsum_slab_rows, int* colsum_rows_out, const int* skip_regions,
                                    int n_regions, float* skip_sum, float* skip_slab, szn_stream_t stream);
int szn_conv2d_wgrad_cb_region(const szn_conv_desc_t* d, int region[8]);

/* ---- upscore: ConvTranspose2d(E,E,64,stride 32,bias=False) with the fixed bilinear kernel of
 * get_upsampling_weight (models.py:11-24,94,146) fused with the crop [19:19+H] (models.py:147).
 * coarse is NHWC f32 [B][h][w] with pixel stride ldc, channels [c0, c0+E); score is NCHW f32.     */
int szn_bilinear_up32_crop_fwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                               const float* coarse, float* score_nchw, szn_stream_t stream);
/* dcoarse[b][i][j][c0+c] = sum_{y,x} dscore[b][c][y][x] * filt[y+crop-32i][x+crop-32j]            */
int szn_bilinear_up32_crop_bwd(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                               const float* dscore_nchw, float* dcoarse, szn_stream_t stream);

/* ---- the same fixed bilinear ConvTranspose2d(E,E,2*stride,stride) + crop for stride 32 (== the two entries above) and
 * stride 8: the last stage of the FCN8s skip head (BASELINE north_star / SURVEY N1: public pytorch-fcn FCN8s, `upscore8`
 * ConvTranspose2d(E,E,16,stride 8) + crop 31; NOT in /root/reference -- parity unpinned).                              */
int szn_bilinear_up_crop_fwd(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                             const float* coarse, float* score_nchw, szn_stream_t stream);
int szn_bilinear_up_crop_bwd(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop,
                             const float* dscore_nchw, float* dcoarse, szn_stream_t stream);
/* FCN8s `upscore2` / `upscore_pool4`: fixed bilinear ConvTranspose2d(C,C,4,stride 2) between NHWC f32 maps with pixel stride ld:
 * in [B][h][w][ld] -> out [B][2h+2][2w+2][ld], channels [0,C); bwd is its transpose (dout -> din).                         */
int szn_bilinear_up2_nhwc_fwd(int B, int h, int w, int C, int ld, const float* in, float* out, szn_stream_t stream);
int szn_bilinear_up2_nhwc_bwd(int B, int h, int w, int C, int ld, const float* dout, float* din, szn_stream_t stream);

/* ---- seenmask_upscore: ConvTranspose2d(C,C,64,stride 32,bias=False) with a LEARNED dense kernel
 * (models.py:98,150-151; trained in phase 2, train.py:170-175).  weight is torch layout
 * (Cin,Cout,64,64) f32; C <= 4.                                                                   */
int szn_deconv64s32_fwd(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                        const float* coarse, const float* weight, float* out_nchw,
                        szn_stream_t stream);
int szn_deconv64s32_dgrad(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                          const float* dout_nchw, const float* weight, float* dcoarse,
                          szn_stream_t stream);
int szn_deconv64s32_wgrad(int B, int h, int w, int C, int ldc, int c0, int H, int W, int crop,
                          const float* coarse, const float* dout_nchw, float* dweight,
                          int accumulate, szn_stream_t stream);

/* ---- phase 2 (BASELINE configs[2]): the seen-mask head fused from the 1/32 map -------------------------
 * Equivalent to szn_deconv64s32_fwd -> [target = np.in1d(label, seen), trainer_seenmask.py:55-56] -> szn_ce2d_fwd
 * (size_average) -> channel argmax (trainer_seenmask.py:67) -> szn_ce2d_bwd -> szn_deconv64s32_dgrad / _wgrad
 * (models.py:150-151, utils.py:19-48) without the (B,2,H,W) score or its gradient in HBM.  coarse: NHWC f32
 * [B][h][w] with pixel stride ldc, the two seen-mask channels at [c0, c0+2); weight: seenmask_upscore.weight
 * (2,2,64,64) f32.  Binary target of a pixel: n_class > 0: (0 <= label < n_class and bit `label` of seen_bits) ? 1 : 0
 * -- unlabelled pixels (-1) become 0 and COUNT, like the reference; labels below -1 mark batch padding (datasets.pad_collate
 * writes -2 where a smaller image of a ragged batch was extended: no counterpart in the reference, which trains at batch size 1)
 * and are ignored like cross_entropy2d ignores negative targets; n_class == 0: `target` already holds {0,1}
 * (anything else is ignored, cross_entropy2d's mask).  Outputs: loss[1]; stats[2] = {sum of terms, valid pixels}
 * (may be NULL); conf[4] int64 += confusion counts [target][prediction] (may be NULL; the running train metrics,
 * trainer_seenmask.py:87); pred int64 (B,H,W) (may be NULL); dscore2 f32 [B*h*w][2] = d loss / d coarse (compact) and
 * dweight (2,2,64,64) = d loss / d weight -- both NULL for a forward-only call.  Bit-reproducible (fixed-order slabs,
 * no atomics).  workspace: szn_seenmask_head_workspace_bytes.                                                      */
size_t szn_seenmask_head_workspace_bytes(int B, int h, int w, int H, int W, int crop);
int szn_seenmask_head(int B, int h, int w, int ldc, int c0, int H, int W, int crop, const float* coarse,
                      const float* weight, const int64_t* target, int n_class, uint64_t seen_bits, float* loss,
                      float* stats, int64_t* conf, int64_t* pred, float* dscore2, float* dweight, void* workspace,
                      szn_stream_t stream);
/* the same with n_class <= SZN_MAX_CLASSES and the seen classes as a szn_class_set */
int szn_seenmask_head_k(int B, int h, int w, int ldc, int c0, int H, int W, int crop, const float* coarse,
                        const float* weight, const int64_t* target, int n_class, const szn_class_set* seen, float* loss,
                        float* stats, int64_t* conf, int64_t* pred, float* dscore2, float* dweight, void* workspace,
                        szn_stream_t stream);
/* seenmask_score = Conv2d(4096, 2, 1) (models.py:97,149) backward from the compact gradient above:
 * dw[c][k] = sum_m dscore2[m][c] * feat[m][k], db[c] = sum_m dscore2[m][c]; feat [M][ldf] of `dtype` (relu7 after
 * Dropout2d), F a multiple of 8.  Fixed-order slabs in `workspace` (szn_seenmask_score_wgrad_workspace_bytes).     */
size_t szn_seenmask_score_wgrad_workspace_bytes(long M, int F);
int szn_seenmask_score_wgrad(int dtype, long M, int F, int ldf, const void* feat, const float* dscore2, float* dw,
                             float* db, void* workspace, szn_stream_t stream);

/* ---- losses (utils.py) --------------------------------------------------------------------------
 * score (B,E,H,W) NCHW f32; target (B,H,W) int64, <0 = ignore.  The target embedding of a pixel is
 * either gathered from embed[K][E] by label (target_embed == NULL; ignore pixels use row 0 exactly
 * like context_dataset.py:128-141) or read from target_embed (B,E,H,W) NCHW f32.
 * Batched definition (the reference is n = 1 only): per-image loss, mean over images.
 * stats: f32 [B][2] = {sum over valid px of the per-pixel term, number of valid px};
 * loss: f32 [1].  workspace: szn_loss_workspace_bytes(B,H,W).                                       */
size_t szn_loss_workspace_bytes(int B, int H, int W);
/* utils.py:75-102  loss_b = (N_b - sum cos) / N_b                                                  */
int szn_cosine_loss_fwd(int B, int E, int H, int W, int K, const float* score,
                        const int64_t* target, const float* embed, const float* target_embed,
                        float* loss, float* stats, void* workspace, szn_stream_t stream);
/* dscore = gout * dloss/dscore  (gout: device scalar, NULL = 1)                                    */
int szn_cosine_loss_bwd(int B, int E, int H, int W, int K, const float* score,
                        const int64_t* target, const float* embed, const float* target_embed,
                        const float* stats, const float* gout, float* dscore, szn_stream_t stream);
/* utils.py:50-73  loss_b = sum_{valid px, c} (s - t)^2 / N_b                                       */
int szn_mse_loss_fwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                     const float* embed, const float* target_embed, float* loss, float* stats,
                     void* workspace, szn_stream_t stream);
int szn_mse_loss_bwd(int B, int E, int H, int W, int K, const float* score, const int64_t* target,
                     const float* embed, const float* target_embed, const float* stats,
                     const float* gout, float* dscore, szn_stream_t stream);
/* utils.py:19-48  cross_entropy2d: sum over ALL valid pixels of the batch of -weight[target] * log_softmax[target]
 * (weight: optional f32 [C] class weights, NULL = 1; F.nll_loss(weight=, size_average=False), utils.py:46);
 * size_average divides by the NUMBER of valid pixels (utils.py:47-48, not by the weight sum).  pred (may be NULL) receives
 * the channel argmax (score.data.max(1)[1], trainer_fcn.py:117 / trainer_seenmask.py:67) as int64 (B,H,W).           */
int szn_ce2d_fwd(int B, int C, int H, int W, const float* score, const int64_t* target, const float* weight,
                 int size_average, float* loss, float* stats, int64_t* pred, void* workspace, szn_stream_t stream);
int szn_ce2d_bwd(int B, int C, int H, int W, const float* score, const int64_t* target, const float* weight,
                 int size_average, const float* stats, const float* gout, float* dscore, szn_stream_t stream);

/* ---- nearest-class-embedding inference (utils.py:159-205) --------------------------------------
 * sim[k] = (score_px . embed[k]) / (||score_px|| * (||embed[k]|| == 0 ? 1 : ||embed[k]||)),
 * pred = first argmax_k.  embed[K][E] f32.  K <= 64 through the uint64_t entry points below, K <= SZN_MAX_CLASSES (256)
 * through the _k forms, which take class sets as szn_class_set (the reference's unseen lists are plain Python lists of any
 * length, trainer_fcn.py:56-64; 64 classes cover its two datasets -- 21 and 33/59 classes -- but not a 150-class label set).
 * mode 0 (infer_lbl):    all K rows of embed compete.
 * mode 1 (stich_seen_unseen_with_mask / infer_lbl_szn / infer_lbl_forced_unseen):
 *        rows with bit k of unseen_bits set form the "unseen-only" matrix, the others the
 *        "seen-only" matrix (the other group's rows zeroed, trainer_fcn.py:56-64, which score
 *        exactly 0 and still compete); a pixel takes the unseen-only prediction when
 *          - seenmask != NULL: argmax over the 2 channels of seenmask (B,2,H,W) is 0
 *            (utils.py:197-198), or
 *          - seenmask == NULL: its target label is an unseen class (utils.py:190-191).
 * pred: int64 (B,H,W).                                                                             */
int szn_embed_argmax(int B, int E, int H, int W, int K, const float* score, const float* embed,
                     int mode, uint64_t unseen_bits, const float* seenmask, const int64_t* target,
                     int64_t* pred, szn_stream_t stream);
int szn_embed_argmax_k(int B, int E, int H, int W, int K, const float* score, const float* embed,
                       int mode, const szn_class_set* unseen, const float* seenmask, const int64_t* target,
                       int64_t* pred, szn_stream_t stream);

/* ---- confusion histogram (utils.py:104-154) ------------------------------------------------------
 * hist[3][K][K] int64 += bincount(K*gt+pred) over pixels with 0 <= gt < K, for {all, gt in seen,
 * gt in unseen}; unseen_bits == 0 fills only hist[0].                                              */
int szn_confusion_hist(long npix, int K, const int64_t* label_true, const int64_t* label_pred,
                       uint64_t unseen_bits, int64_t* hist, szn_stream_t stream);
/* K <= SZN_MAX_CLASSES; an empty or NULL set fills only hist[0] */
int szn_confusion_hist_k(long npix, int K, const int64_t* label_true, const int64_t* label_pred,
                         const szn_class_set* unseen, int64_t* hist, szn_stream_t stream);

/* ---- fused head: coarse -> (loss, prediction, dcoarse) without materialising (B,E,H,W) ----------
 * Equivalent to szn_bilinear_up32_crop_fwd -> szn_cosine_loss_fwd -> szn_embed_argmax ->
 * szn_cosine_loss_bwd -> szn_bilinear_up32_crop_bwd (models.py:146-147 + utils.py:75-102,159-185)
 * evaluated algebraically per 32x32 cell.  pred / dcoarse may be NULL.  dcoarse is written in
 * `dcoarse_dtype` with pixel stride ldc (channels [c0,c0+E) only).                                 */
size_t szn_fused_head_workspace_bytes(int B, int h, int w, int E, int K);
int szn_fused_head(int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                   const float* coarse, const float* embed, const int64_t* target,
                   float* loss, float* stats, int64_t* pred, int dcoarse_dtype, void* dcoarse,
                   void* workspace, szn_stream_t stream);
/* the same head over stride-8 cells (coarse = the 1/8 fused map of the FCN8s skip head, crop 31); stride in {32, 8};
 * szn_fused_head == stride 32.  The workspace size is szn_fused_head_workspace_bytes of the map's own h, w.            */
int szn_fused_head_strided(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                           const float* coarse, const float* embed, const int64_t* target, float* loss, float* stats,
                           int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace, szn_stream_t stream);
/* The class embeddings are constants of a run (trainer_fcn.py:49-62): szn_fused_head_prepare writes their transpose and norms to the head of
 * `workspace` once, szn_fused_head_prepared is szn_fused_head_strided without that launch (23 us of dependent loads per step) for a caller
 * that keeps the workspace and prepares again whenever the embeddings or the workspace change.  Same results bit for bit. */
int szn_fused_head_prepare(int E, int K, const float* embed, void* workspace, szn_stream_t stream);
int szn_fused_head_prepared(int stride, int B, int h, int w, int E, int ldc, int c0, int H, int W, int crop, int K,
                            const float* coarse, const float* embed, const int64_t* target, float* loss, float* stats,
                            int64_t* pred, int dcoarse_dtype, void* dcoarse, void* workspace, szn_stream_t stream);

/* ---- optimizers (train.py:126-133,174-175; torch.optim.Adam / SGD semantics) ----------------------
 * One launch per flat fp32 parameter buffer.  grad_scale multiplies the gradient first (1/world
 * after a sum all-reduce; 1/(world * loss_scale) on the fp16 path).  If w_lp != NULL the updated weight is
 * also written as a 16-bit image of type w_lp_dtype (SZN_BF16 | SZN_F16) for the next forward pass.    */
int szn_adam_step(long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                  float grad_scale, void* w_lp, int w_lp_dtype, szn_stream_t stream);
int szn_sgd_momentum_step(long n, float* param, const float* grad, float* momentum_buf, float lr,
                          float momentum, float weight_decay, int first_step, float grad_scale,
                          void* w_lp, int w_lp_dtype, szn_stream_t stream);
/* The same steps reading the gradient as a 16-bit image (grad_dtype = SZN_BF16 | SZN_F16) -- the summed wire buffer of the
 * data-parallel exchange (szn_conv_desc_t.dw_lp -> all-reduce / reduce-scatter in place -> here): the widening copy back into an
 * fp32 gradient buffer disappears and the pass reads 2 B per weight instead of 4.  Arithmetic after widening: identical.   */
int szn_adam_step_g16(long n, float* param, const void* grad, int grad_dtype, float* exp_avg, float* exp_avg_sq,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      float grad_scale, void* w_lp, int w_lp_dtype, szn_stream_t stream);
int szn_sgd_momentum_step_g16(long n, float* param, const void* grad, int grad_dtype, float* momentum_buf, float lr,
                              float momentum, float weight_decay, int first_step, float grad_scale,
                              void* w_lp, int w_lp_dtype, szn_stream_t stream);

/* ---- dynamic loss scaling for the IEEE-half path (BASELINE configs[4]: "fp16 activations"; the reference is fp32 and has
 * no counterpart).  scale_state: device float[4] = {loss scale S, found_inf flag, optimizer steps applied, clean steps
 * since S last changed}.  Per step: d(loss)/d(coarse) is multiplied by S in fp32 before it enters the 16-bit backward pass;
 * szn_grad_check_finite (after the gradient all-reduce, so every rank sees the same flag: inf / NaN survive a sum) raises
 * found_inf if any gradient element is not finite; the *_scaled optimizer entry points read S, the flag and the step count
 * from scale_state: they divide the gradient by S and do NOTHING when the flag is set (masters, moments and the 16-bit
 * weight image stay as they were); szn_loss_scale_update then backs S off (x backoff, >= min_scale) after an overflow,
 * or counts the step and grows S (x growth, <= max_scale) after growth_interval clean steps, and clears the flag.
 * No host synchronisation anywhere: the scale never leaves the device.                                                  */
int szn_grad_check_finite(long n, const float* grad, float* scale_state, szn_stream_t stream);
int szn_adam_step_scaled(long n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                         float beta2, float eps, float weight_decay, const float* scale_state, float grad_scale,
                         void* w_lp, int w_lp_dtype, szn_stream_t stream);
int szn_sgd_momentum_step_scaled(long n, float* param, const float* grad, float* momentum_buf, float lr, float momentum,
                                 float weight_decay, const float* scale_state, float grad_scale, void* w_lp,
                                 int w_lp_dtype, szn_stream_t stream);
int szn_loss_scale_update(float* scale_state, float growth, float backoff, int growth_interval, float min_scale,
                          float max_scale, szn_stream_t stream);

/* ---- row / column remapping of an NHWC map (the constant band of the pad-100 network inside the conv3 block) --------------
 * models.py:43 pads conv1_1 by 100: at 1/4 resolution most rows / columns between the tensor edge and the image's reach hold one
 * value per channel, and conv3_1 .. conv3_3 (models.py:56-62,123-128) need not compute them.  One kernel does the four index maps
 * the caller needs (crop, put the pooled rows back, and their two backward forms):
 *     out[b][y][x][c] = sum over sy in [ytab[2y], ytab[2y] + ytab[2y+1]), sx in [xtab[2x], xtab[2x] + xtab[2x+1]) of in[b][sy][sx][c]
 * (count 0 = zeros; fp32 accumulation in ascending order; a single source is moved bit for bit).  in [B][Hi][Wi][C], out
 * [B][Ho][Wo][C] dense, 16-B aligned, C a multiple of 8 (16-bit) / 4 (f32); ytab [Ho][2], xtab [Wo][2] int32 on the device.          */
int szn_band_remap(int dtype, int B, int Hi, int Wi, int Ho, int Wo, int C, const void* in, void* out, const int* ytab,
                   const int* xtab, szn_stream_t stream);
/* In place: every run {start, count} of `runs` (device int [n_runs][2]) summed (fp32, ascending) into its first row (axis 0) / column (axis 1) of
 * d [B][H][W][C].  The summing rows / columns of a transposed band map are few; folded first, the map is a one-source gather that
 * szn_maxpool2x2_ceil_bwd_code_gather applies while reading (rows, then columns = the two szn_band_remap passes, bit for bit). */
int szn_band_fold(int dtype, int B, int H, int W, int C, void* d, int axis, const int* runs, int n_runs, szn_stream_t stream);

/* ---- small utilities -------------------------------------------------------------------------------- */
int szn_cast(int src_dtype, int dst_dtype, long n, const void* src, void* dst, szn_stream_t stream);
/* Dropout2d factors: scale[i] = (u_i >= p) ? 1/(1-p) : 0 with a counter-based generator (seed, i)   */
int szn_dropout2d_mask(long n, float p, uint64_t seed, uint64_t offset, float* scale,
                       szn_stream_t stream);
/* fp8 pixel projection (BASELINE configs[4]: "fp8 MFMA projection GEMM"; score_fr || seenmask_score forward, models.py:
 * 93,97,145,149).  x [M][K] (x_dtype) and w [N][K] (w_dtype) are quantised to OCP e4m3 with per-tensor scales
 * amax/448 (round to nearest even), multiplied on the fp8 matrix cores with fp32 accumulation and rescaled:
 * out[m][n] = (sum_k xq wq) * sx * sw + bias[n], fp32, row stride ldo.  K must be a multiple of 128.  workspace:
 * szn_proj_fp8_workspace_bytes(M, K, N) bytes, 16-B aligned.                                                       */
size_t szn_proj_fp8_workspace_bytes(long M, int K, int N);
/* Backward of that layer on the fp8 matrix cores (the e4m3-forward / e5m2-gradient recipe; no reference counterpart):
 *   dgrad: dx[m][k] = epi( sum_n g[m][n] * w[n][k] ),  wgrad: dw[n][k] = sum_m g[m][n] * x[m][k]
 * g (the gradient wrt the projection output, row stride ldg) is quantised to OCP e5m2 with the per-tensor scale
 * amax / 57344, w [N][K] and x [M][ldx] to e4m3 (amax / 448); products exact in fp32, fp32 accumulation, one rescale.
 * dgrad epilogue like szn_conv2d_dgrad: optional gate (gate[m][k] > 0 ? v : 0; element type gate_dtype, row stride ldgate),
 * optional chan_scale[image][k] (image = m / rows_per_image), output element type out_dtype, row stride ldx.
 * dw is fp32 [N][K], dense.  workspace: szn_proj_fp8_bwd_workspace_bytes(M, K, N), 16-B aligned.                      */
size_t szn_proj_fp8_bwd_workspace_bytes(long M, int K, int N);
int szn_proj_fp8_dgrad(int g_dtype, int w_dtype, long M, int K, int N, int ldg, const void* g, const void* w,
                       const void* gate, int gate_dtype, int ldgate, const float* chan_scale, int rows_per_image,
                       int out_dtype, void* dx, int ldx, void* workspace, szn_stream_t stream);
int szn_proj_fp8_wgrad(int g_dtype, int x_dtype, long M, int K, int N, int ldg, int ldx, const void* g, const void* x,
                       float* dw, void* workspace, szn_stream_t stream);
int szn_proj_fp8_fwd(int x_dtype, int w_dtype, long M, int K, int N, int ldo, const void* x, const void* w,
                     const float* bias, float* out_f32, void* workspace, szn_stream_t stream);
/* Dataset transform on the device (context_dataset.py:143-150, pascal_dataset.py:138-145): RGB uint8 HWC image(s)
 * [B][H][W][3] -> BGR, minus mean_bgr (three doubles, BGR order), as the (B,3,H,W) f32 NCHW network input.  The
 * subtraction is done in float64 and rounded once to float32, like the reference (bit-identical).              */
int szn_image_u8_to_bgr_f32(int B, int H, int W, const uint8_t* rgb_hwc, const double* mean_bgr, float* out_nchw,
                            szn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SZN_H_ */
