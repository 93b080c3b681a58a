// edtts_melpost.h -- mel post-processing kernels: the step right after the sampler in the
// reference's scripts (generate_sample.py:115-145, inference_pipeline.py:382-396): denormalize_mel (utils/audio.py:17-19) -> exp
// -> torchaudio InverseMelScale (minimum-norm least squares = multiplication by the pseudo-inverse of the mel filter bank, relu)
// -> torchaudio GriffinLim (n_iter x [istft -> stft -> momentum phase update], final istft).  Included by edtts_kernels.hip.
//
// Bounds: the inverse-mel product is a small GEMM ([513 x 80] per frame: 82 kFLOP/frame, HBM traffic 320 B in / 2 KB out per
// frame); Griffin-Lim is FFT-bound: per iteration and frame one 1024-point inverse and one forward complex FFT (2 x 51 kFLOP)
// in LDS plus 4 KB (frame buffer) + 8 KB (phase state) + 4 KB (spectrum) of HBM traffic -> ~25 FLOP/B: HBM-bound on MI355X
// (ridge 20 FLOP/B for fp32 vector math); the iteration state is laid out frame-major so that every access is a contiguous row.
#pragma once

namespace melpost {
using namespace edtts;

constexpr int kNfft = 1024, kBins = kNfft / 2 + 1, kThreads = 256;

struct cplx {
  float re, im;
};
EDTTS_DEV cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }

// 1024-point complex FFT in LDS, radix-2 Stockham autosort (natural order in and out), 256 threads x 2 butterflies per stage.
// tw[q] = exp(-2 pi i q / 1024), q < 512 (host table, evaluated in fp64); INV conjugates it (no 1/N scaling here).
// x holds the input; returns the buffer (x or y) that holds the result.  Callers synchronise before reading.
template <bool INV>
EDTTS_DEV cplx* fft1024(cplx* x, cplx* y, const cplx* __restrict__ tw, int tid) {
  constexpr int N = kNfft, T = N / 2;
#pragma unroll 1
  for (int p = 1; p < N; p <<= 1) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * kThreads;      // butterfly index in [0, 512)
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      cplx w = tw[k * (T / p)];              // exp(-i pi k / p)
      if (INV) w.im = -w.im;
      const cplx u0 = x[i], u1 = cmul(w, x[i + T]);
      y[j] = {u0.re + u1.re, u0.im + u1.im};
      y[j + p] = {u0.re - u1.re, u0.im - u1.im};
    }
    cplx* t = x; x = y; y = t;
  }
  __syncthreads();
  return x;
}

// spec[b][f][t] = relu(sum_m pinv[f][m] * lin[b][t][m]),  lin = exp(mel_n * std + mean) or the input itself   (torch layout [B, n_freqs, T])
__global__ __launch_bounds__(kThreads) void k_mel_to_spec(const float* __restrict__ mel, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          const float* __restrict__ pinv, int T, int M, int NFQ, float* __restrict__ spec) {
  extern __shared__ float lin[];  // [16 frames][M]
  const int b = blockIdx.y, t0 = blockIdx.x * 16;
  for (int i = threadIdx.x; i < 16 * M; i += kThreads) {
    const int tl = i / M, m = i % M, t = t0 + tl;
    float v = 0.f;
    if (t < T) {
      v = mel[((size_t)b * T + t) * M + m];
      // normalised log-mel in (mean / std given): denormalise (utils/audio.py:19) and leave the log domain (generate_sample.py:119);
      // otherwise the input already is the linear mel spectrogram (plain torchaudio InverseMelScale semantics)
      if (stdv) v = expf(v * stdv[(size_t)b * M + m] + mean[(size_t)b * M + m]);
    }
    lin[i] = v;
  }
  __syncthreads();
  // thread -> (frequency f, 16 frames): the 16 outputs of a frequency are contiguous in the [.., f, t] layout
  for (int f = threadIdx.x; f < NFQ; f += kThreads) {
    float acc[16];
#pragma unroll
    for (int tl = 0; tl < 16; ++tl) acc[tl] = 0.f;
    const float* pr = pinv + (size_t)f * M;
    for (int m = 0; m < M; ++m) {
      const float w = pr[m];
#pragma unroll
      for (int tl = 0; tl < 16; ++tl) acc[tl] = fmaf(w, lin[tl * M + m], acc[tl]);
    }
    float* o = spec + ((size_t)b * NFQ + f) * T + t0;
#pragma unroll
    for (int tl = 0; tl < 16; ++tl)
      if (t0 + tl < T) o[tl] = fmaxf(acc[tl], 0.f);
  }
}

// mag[b][t][f] = spec[b][f][t] ^ (1 / power); angles[b][t][f] = angles0 (torch layout [B, n_freqs, T] complex) or uniform draws
__global__ __launch_bounds__(kThreads) void k_gl_init(const float* __restrict__ spec, const float* __restrict__ angles0, int T, float inv_power,
                                                      unsigned long long seed, float* __restrict__ mag, cplx* __restrict__ ang, cplx* __restrict__ tprev) {
  const int b = blockIdx.y, t = blockIdx.x;
  for (int f = threadIdx.x; f < kBins; f += kThreads) {
    const size_t src = ((size_t)b * kBins + f) * T + t, dst = ((size_t)b * T + t) * kBins + f;
    const float s = spec[src];
    mag[dst] = inv_power == 0.5f ? sqrtf(s) : powf(s, inv_power);
    cplx a;
    if (angles0) a = {angles0[2 * src], angles0[2 * src + 1]};
    else {  // torch.rand(complex): real and imaginary parts uniform in [0, 1)
      // counter hash (splitmix64 of (seed, element)) -> two 24-bit uniforms; the reference's draw comes from torch's global RNG
      unsigned long long z = (seed ^ 0x9E3779B97F4A7C15ull) + (unsigned long long)src * 0xBF58476D1CE4E5B9ull;
      z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
      a = {(float)(unsigned)(z >> 40) * (1.0f / 16777216.0f), (float)(unsigned)((z >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f)};
    }
    ang[dst] = a;
    tprev[dst] = {0.f, 0.f};
  }
}

// frames[b][t][n] = window[n] * irfft(mag * angles)[n]      (torch.istft: per-frame inverse real FFT, "backward" normalisation)
__global__ __launch_bounds__(kThreads) void k_gl_istft(const float* __restrict__ mag, const cplx* __restrict__ ang, const float* __restrict__ window,
                                                       const cplx* __restrict__ tw, int T, float* __restrict__ frames) {
  __shared__ cplx bufa[kNfft], bufb[kNfft];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const size_t row = ((size_t)b * T + t) * kBins;
  for (int f = tid; f < kBins; f += kThreads) {
    const float m = mag[row + f];
    cplx v = {m * ang[row + f].re, m * ang[row + f].im};
    if (f == 0 || f == kNfft / 2) v.im = 0.f;  // a real signal's DC / Nyquist bins (irfft ignores their imaginary parts)
    bufa[f] = v;
    if (f > 0 && f < kNfft / 2) bufa[kNfft - f] = {v.re, -v.im};  // Hermitian extension
  }
  cplx* r = fft1024<true>(bufa, bufb, tw, tid);
  float* out = frames + ((size_t)b * T + t) * kNfft;
  for (int n = tid; n < kNfft; n += kThreads) out[n] = r[n].re * (1.0f / kNfft) * window[n];
}

// wave[b][p] = sum_t frames[b][t][p - t hop] / sum_t window^2[p - t hop]   over the frames that cover padded position p
// (torch.istft overlap-add and window-envelope normalisation; ascending t: deterministic)
__global__ __launch_bounds__(kThreads) void k_gl_ola(const float* __restrict__ frames, const float* __restrict__ window, int T, int hop, int Lp,
                                                     float* __restrict__ wave) {
  const int b = blockIdx.y;
  for (int p = blockIdx.x * kThreads + threadIdx.x; p < Lp; p += gridDim.x * kThreads) {
    int t_lo = (p - (kNfft - 1) + hop - 1) / hop;
    if (p - (kNfft - 1) < 0) t_lo = 0;
    int t_hi = p / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    float s = 0.f, e = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int n = p - t * hop;
      s += frames[((size_t)b * T + t) * kNfft + n];
      const float w = window[n];
      e = fmaf(w, w, e);
    }
    wave[(size_t)b * Lp + p] = e > 1e-11f ? s / e : 0.f;
  }
}

// rebuilt = stft(x, center=True, reflect), x = wave[n_fft/2 : n_fft/2 + Lx];  then the momentum phase update
//   a = rebuilt - mom * tprev;  angles = a / (|a| + 1e-16);  tprev = rebuilt          (torchaudio.functional.griffinlim)
__global__ __launch_bounds__(kThreads) void k_gl_stft(const float* __restrict__ wave, const float* __restrict__ window, const cplx* __restrict__ tw,
                                                      int T, int hop, int Lp, float mom, cplx* __restrict__ ang, cplx* __restrict__ tprev) {
  __shared__ cplx bufa[kNfft], bufb[kNfft];
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const int Lx = Lp - kNfft;  // = hop * (T - 1)
  const float* x = wave + (size_t)b * Lp + kNfft / 2;
  for (int n = tid; n < kNfft; n += kThreads) {
    int idx = t * hop + n - kNfft / 2;
    if (idx < 0) idx = -idx;
    if (idx >= Lx) idx = 2 * (Lx - 1) - idx;
    idx = idx < 0 ? 0 : idx;  // (signals shorter than the pad are not supported by torch either)
    bufa[n] = {x[idx] * window[n], 0.f};
  }
  cplx* r = fft1024<false>(bufa, bufb, tw, tid);
  const size_t row = ((size_t)b * T + t) * kBins;
  for (int f = tid; f < kBins; f += kThreads) {
    const cplx rb = r[f], tp = tprev[row + f];
    const cplx a = {rb.re - mom * tp.re, rb.im - mom * tp.im};
    const float inv = 1.0f / (sqrtf(a.re * a.re + a.im * a.im) + 1e-16f);
    ang[row + f] = {a.re * inv, a.im * inv};
    tprev[row + f] = rb;
  }
}

}  // namespace melpost
