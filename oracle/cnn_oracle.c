/* ORACLE (test infrastructure; never linked into or called by the product).
 *
 * Plain-C restatement of the tensor operators the reference's two CNN graphs
 * invoke through tch 0.3.0 -> libtorch 1.7.0 (not vendored in /root/reference):
 *   text_detection/model.rs:107-151  conv2d(bias=false), batch_norm (eval),
 *       relu, max_pool2d(3,2,1), upsample_nearest2d, add, cat, conv_transpose2d
 *       (k=2,s=2,bias), sigmoid
 *   char_recognition/model.rs:28-39  conv2d(+bias), max_pool2d(2), linear, relu
 *   char_recognition/mod.rs:53-56, utils.rs:28-43  softmax(-1, Double) + top-1
 * Layout NCHW contiguous f32, exactly the tensors tch hands to ATen.  Every
 * output element is one sequential f32 accumulation in (ci, kh, kw) order.
 * The graphs themselves are composed in oracle/cnn_oracle.py.
 *
 * PARITY UNPINNED by the reference (no weights, no expected activations in any
 * of its tests); cross-checked against ATen via oracle/torch_ref.py.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#define IDX4(n, c, h, w, C, H, W) ((((size_t)(n) * (C) + (c)) * (H) + (h)) * (W) + (w))

void orc_conv2d(const float* x, int N, int C, int H, int W, const float* wt, int Co, int K,
                int stride, int pad, const float* bias, float* y) {
  int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int oh = 0; oh < Ho; ++oh)
        for (int ow = 0; ow < Wo; ++ow) {
          float acc = 0.f;
          for (int ci = 0; ci < C; ++ci)
            for (int kh = 0; kh < K; ++kh) {
              int ih = oh * stride - pad + kh;
              if (ih < 0 || ih >= H) continue;
              for (int kw = 0; kw < K; ++kw) {
                int iw = ow * stride - pad + kw;
                if (iw < 0 || iw >= W) continue;
                acc += x[IDX4(n, ci, ih, iw, C, H, W)] * wt[IDX4(co, ci, kh, kw, C, K, K)];
              }
            }
          if (bias) acc += bias[co];
          y[IDX4(n, co, oh, ow, Co, Ho, Wo)] = acc;
        }
}

/* eval-mode batch norm, eps as given (tch default 1e-5) */
void orc_batch_norm(float* x, int N, int C, int HW, const float* gamma, const float* beta,
                    const float* mean, const float* var, float eps) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c) {
      float invstd = 1.0f / sqrtf(var[c] + eps);
      float* p = x + ((size_t)n * C + c) * HW;
      for (int i = 0; i < HW; ++i) p[i] = (p[i] - mean[c]) * invstd * gamma[c] + beta[c];
    }
}

void orc_relu(float* x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

void orc_add(float* x, const float* y, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = x[i] + y[i];
}

void orc_max_pool2d(const float* x, int N, int C, int H, int W, int K, int stride, int pad, float* y) {
  int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
  for (int nc = 0; nc < N * C; ++nc)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        float m = -INFINITY;
        for (int kh = 0; kh < K; ++kh) {
          int ih = oh * stride - pad + kh;
          if (ih < 0 || ih >= H) continue;
          for (int kw = 0; kw < K; ++kw) {
            int iw = ow * stride - pad + kw;
            if (iw < 0 || iw >= W) continue;
            float v = x[((size_t)nc * H + ih) * W + iw];
            if (v > m) m = v;
          }
        }
        y[((size_t)nc * Ho + oh) * Wo + ow] = m;
      }
}

/* upsample_nearest2d to exactly k x size: dst[y][x] = src[y/k][x/k] */
void orc_upsample_nearest(const float* x, int NC, int H, int W, int k, float* y) {
  int Ho = H * k, Wo = W * k;
  for (int nc = 0; nc < NC; ++nc)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        y[((size_t)nc * Ho + oh) * Wo + ow] = x[((size_t)nc * H + oh / k) * W + ow / k];
}

/* conv_transpose2d k=2 s=2 p=0, weight [Cin, Cout, 2, 2], bias [Cout] */
void orc_conv_transpose2d_k2s2(const float* x, int N, int C, int H, int W, const float* wt, int Co,
                               const float* bias, float* y) {
  int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Co; ++co)
      for (int oh = 0; oh < Ho; ++oh)
        for (int ow = 0; ow < Wo; ++ow) {
          int i = oh / 2, a = oh % 2, j = ow / 2, b = ow % 2;
          float acc = 0.f;
          for (int ci = 0; ci < C; ++ci)
            acc += x[IDX4(n, ci, i, j, C, H, W)] * wt[IDX4(ci, co, a, b, Co, 2, 2)];
          y[IDX4(n, co, oh, ow, Co, Ho, Wo)] = acc + (bias ? bias[co] : 0.f);
        }
}

void orc_sigmoid(float* x, size_t n) {
  for (size_t i = 0; i < n; ++i) x[i] = 1.0f / (1.0f + expf(-x[i]));
}

/* linear: y[n][o] = sum_i x[n][i] * w[o][i] + b[o] */
void orc_linear(const float* x, int N, int I, const float* w, int O, const float* b, float* y) {
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < O; ++o) {
      float acc = 0.f;
      for (int i = 0; i < I; ++i) acc += x[(size_t)n * I + i] * w[(size_t)o * I + i];
      y[(size_t)n * O + o] = acc + b[o];
    }
}

/* softmax(-1, Kind::Double) then topk(1): label index and its f64 probability */
void orc_softmax_top1(const float* logits, int N, int C, int32_t* label, double* prob) {
  for (int n = 0; n < N; ++n) {
    const float* l = logits + (size_t)n * C;
    double mx = (double)l[0];
    int arg = 0;
    for (int c = 1; c < C; ++c)
      if ((double)l[c] > mx) { mx = (double)l[c]; arg = c; }
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += exp((double)l[c] - mx);
    label[n] = arg;
    prob[n] = 1.0 / s; /* exp(mx - mx) / s */
  }
}
