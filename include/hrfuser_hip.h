/* hrfuser_hip.h — C ABI of libhrfuser_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the HRFuser backbone hot path.  The reference has no
 * native layer: its backbone calls PyTorch ATen ops (cuDNN/cuBLAS underneath).  Every entry
 * point below names the ATen call sites in /root/reference it replaces; a host binding
 * (ctypes here, cgo/JNI elsewhere) passes raw device pointers, sizes and a hipStream_t.
 *
 * Conventions
 *   - all activations fp32, NHWC ("NLC" in the reference) unless explicit element strides
 *     (sB,sY,sX,sC) are taken, which also admit NCHW views (network inputs / input grads);
 *   - weights keep the reference's parameter layouts (Conv2d OIHW, Linear (out,in)) so
 *     state-dicts load unchanged;
 *   - every function enqueues on `stream` and returns immediately: 0 = HRF_OK, 1 = bad argument,
 *     2 = launch failure.  Nothing throws, allocates, or synchronises.  Re-entrant per device.
 *   - "stats" buffers are double[2*C] = (sum, sum of squares | sum, sum*x) accumulated with
 *     atomics: zero them (hipMemsetAsync) before the producing launch.
 *   - transform-on-load modes (tf_mode): 0 none, 1 affine, 2 affine+ReLU, 3 affine+GELU(erf),
 *     4 LayerNorm (rowstat = (rows,2) mean,rstd).  They replace materialised BatchNorm /
 *     LayerNorm / activation tensors of the reference graph.
 */
#ifndef HRFUSER_HIP_H_
#define HRFUSER_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense convolution / Linear engine (fp32 MFMA implicit GEMM) -------------------------
 * Replaces F.conv2d(k=1|3, groups=1) and F.linear:
 *   stems hrnet.py:341-358, hrfuser_hrformer_based.py:380-396; Bottleneck resnet.py:166-205;
 *   transitions hrnet.py:430-459; CrossFFN 1x1 hrformer.py:268,280; fuse 1x1
 *   hrformer.py:511-517,544-551; qkv/out_proj hrformer.py:84,86; q/k/v/out_proj
 *   hrfuser_hrformer_based.py:92-96.  Linear = 1x1 conv over a (1,1,rows,C) view.            */
int hrf_conv_fwd(const float* x, int sB, int sY, int sX, int sC, int B, int H, int W, int Cin,
                 const float* w, const float* bias, int KH, int stride, int Cout,
                 float* y, int ldY, int yoff, const float* res, int ldR,
                 int tf_mode, const float* tf_scale, const float* tf_shift,
                 const float* tf_rowstat, double* stats, void* stream);
/* dX (or, epi=1, dU = dX*act'(scale*xraw+shift) plus (sum dU, sum dU*xraw) for the producer BN).
 * (cA,cB,cC) != NULL applies the BatchNorm backward on load: dy = cA*du + cB*yraw + cC.        */
int hrf_conv_bwd_data(const float* dy, int ldD, int doff, const float* yraw,
                      const float* cA, const float* cB, const float* cC,
                      const float* w, int KH, int stride, int Cout,
                      int B, int H, int W, int Cin,
                      float* dx, int sB, int sY, int sX, int sC, int accumulate,
                      int epi, const float* xraw, int ldXr, const float* tf_scale,
                      const float* tf_shift, int act, double* stats, void* stream);
/* dW += , dbias += (split-K over pixels, fp32 atomics: zero or pre-load the targets).          */
int hrf_conv_bwd_weight(const float* dy, int ldD, int doff, const float* yraw,
                        const float* cA, const float* cB, const float* cC,
                        const float* x, int sB, int sY, int sX, int sC,
                        int B, int H, int W, int Cin, int KH, int stride, int Cout,
                        int tf_mode, const float* tf_scale, const float* tf_shift,
                        const float* tf_rowstat, float* dw, float* dbias, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HRFUSER_HIP_H_ */
