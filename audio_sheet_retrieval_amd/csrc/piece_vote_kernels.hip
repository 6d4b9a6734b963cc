// gfx950 kernels for the piece-identification vote on top of top-k retrieval
// (reference: audio_sheet_server.py:213-300 detect_score / detect_performance - SURVEY.md 8f row 1).
//
//   slice_windows_kernel : n_samples sliding windows cut out of one long spectrogram / unrolled sheet strip
//                          (:217-224, :270-281) into the (n,1,h,w) batch the towers take.  HBM copy.
//   vote_count_kernel    : histogram of the piece ids of the retrieved data-base entries (:228-235 np.unique with
//                          counts) - integer atomics on a per-piece counter array.
//   vote_select_kernel   : the top_k pieces by vote count (:238 argsort(counts)[::-1][:top_k]); one workgroup,
//                          repeated arg-max.  Ties: the LARGER piece id first (what reversing a stable ascending
//                          sort gives; the reference's quicksort leaves ties undefined).
#include "asr_kernels.h"

namespace asr {

__global__ __launch_bounds__(256) void slice_windows_kernel(const float *__restrict__ src, int64_t T, int r0, int win_h,
                                                            int win_w, const int32_t *__restrict__ starts, int n,
                                                            float *__restrict__ out) {
    const int64_t per = (int64_t)win_h * win_w;
    const int64_t total = per * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / per);
        const int rem = (int)(e - (int64_t)i * per);
        const int r = rem / win_w, c = rem - r * win_w;
        out[e] = src[(int64_t)(r0 + r) * T + starts[i] + c];
    }
}

__global__ __launch_bounds__(256) void vote_count_kernel(const int32_t *__restrict__ idx, int64_t n_idx,
                                                         const int32_t *__restrict__ ids, int64_t n_db, int32_t n_pieces,
                                                         int32_t *__restrict__ counts) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_idx; e += (int64_t)gridDim.x * blockDim.x) {
        const int32_t j = idx[e];
        if (j < 0 || j >= n_db) continue;                 // -1: fewer than k data-base entries
        const int32_t p = ids[j];
        if (p >= 0 && p < n_pieces) atomicAdd(&counts[p], 1);
    }
}

// out_piece[r], out_count[r] for r < top_k; unused slots get piece -1 / count 0.  counts is consumed (winners zeroed).
__global__ __launch_bounds__(1024) void vote_select_kernel(int32_t *__restrict__ counts, int32_t n_pieces, int top_k,
                                                           int32_t *__restrict__ out_piece,
                                                           int32_t *__restrict__ out_count) {
    __shared__ int32_t bc[1024], bp[1024];
    const int tid = threadIdx.x;
    for (int r = 0; r < top_k; ++r) {
        int32_t c = 0, p = -1;
        for (int32_t q = tid; q < n_pieces; q += 1024) {
            const int32_t v = counts[q];
            if (v > c || (v == c && v > 0 && q > p)) { c = v; p = q; }
        }
        bc[tid] = c; bp[tid] = p;
        __syncthreads();
        for (int st = 512; st > 0; st >>= 1) {
            if (tid < st) {
                const int32_t c2 = bc[tid + st], p2 = bp[tid + st];
                if (c2 > bc[tid] || (c2 == bc[tid] && c2 > 0 && p2 > bp[tid])) { bc[tid] = c2; bp[tid] = p2; }
            }
            __syncthreads();
        }
        if (tid == 0) {
            out_piece[r] = bc[0] > 0 ? bp[0] : -1;
            out_count[r] = bc[0];
            if (bc[0] > 0) counts[bp[0]] = 0;
        }
        __syncthreads();
    }
}

hipError_t launch_slice_windows(hipStream_t s, const float *src, int64_t T, int r0, int win_h, int win_w,
                                const int32_t *starts_dev, int n, float *out) {
    const int64_t total = (int64_t)win_h * win_w * n;
    if (total == 0) return hipSuccess;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 4096);
    slice_windows_kernel<<<blocks, 256, 0, s>>>(src, T, r0, win_h, win_w, starts_dev, n, out);
    return hipGetLastError();
}

hipError_t launch_piece_vote(hipStream_t s, const int32_t *idx, int64_t n_idx, const int32_t *ids, int64_t n_db,
                             int32_t n_pieces, int top_k, int32_t *counts_ws, int32_t *out_piece, int32_t *out_count) {
    hipError_t e = hipMemsetAsync(counts_ws, 0, (size_t)n_pieces * sizeof(int32_t), s);
    if (e != hipSuccess) return e;
    if (n_idx > 0) {
        const int blocks = (int)std::min<int64_t>((n_idx + 255) / 256, 1024);
        vote_count_kernel<<<blocks, 256, 0, s>>>(idx, n_idx, ids, n_db, n_pieces, counts_ws);
    }
    vote_select_kernel<<<1, 1024, 0, s>>>(counts_ws, n_pieces, top_k, out_piece, out_count);
    return hipGetLastError();
}

}  // namespace asr

// ---- batch assembly of the training pool on the device ------------------------------------
// utils/data_pools.py:127-228 (prepare_train_image / prepare_train_audio / __getitem__): every sample is a window
// of one strip of the resident pool, optionally nearest-neighbour rescaled (cv2.resize INTER_NEAREST: source index
// = min(floor(dst * (1 / (dst_size / src_size))), src_size - 1)) and edge-padded.  The host draws the random
// numbers in the reference's order and boils each sample down to 9 doubles; the kernel is a pure gather:
//   out[i,0,y,x] = src[off + clamp(floor((y0 + y) * sy), 0, ymax) * stride + xadd + clamp(floor((x0 + x) * sx), 0, xmax)]
namespace asr {

__global__ __launch_bounds__(256) void gather_windows_kernel(const float *__restrict__ src, const double *__restrict__ desc,
                                                             int n, int out_h, int out_w, float *__restrict__ out) {
    const int per = out_h * out_w;
    for (int i = blockIdx.y; i < n; i += gridDim.y) {
        const double *d = desc + (size_t)i * 9;
        const int64_t off = (int64_t)d[0], stride = (int64_t)d[1];
        const double y0 = d[2], sy = d[3], ymax = d[4], x0 = d[5], sx = d[6], xmax = d[7];
        const int64_t xadd = (int64_t)d[8];
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per; e += gridDim.x * blockDim.x) {
            const int y = e / out_w, x = e - y * out_w;
            double ry = floor((y0 + (double)y) * sy), rx = floor((x0 + (double)x) * sx);
            ry = ry < 0.0 ? 0.0 : (ry > ymax ? ymax : ry);
            rx = rx < 0.0 ? 0.0 : (rx > xmax ? xmax : rx);
            out[(size_t)i * per + e] = src[off + (int64_t)ry * stride + xadd + (int64_t)rx];
        }
    }
}

hipError_t launch_gather_windows(hipStream_t s, const float *src, const double *desc_dev, int n, int out_h, int out_w,
                                 float *out) {
    if (n == 0 || out_h * out_w == 0) return hipSuccess;
    const int bx = std::max(1, std::min((out_h * out_w + 255) / 256, 32));
    gather_windows_kernel<<<dim3(bx, std::min(n, 4096)), 256, 0, s>>>(src, desc_dev, n, out_h, out_w, out);
    return hipGetLastError();
}

}  // namespace asr

// ---- audio front-end ---------------------------------------------------------------------
// The madmom chain of the reference (tutorials/Embedding Tutorial.ipynb cell 28; msmd.midi_parser.processor):
// FramedSignal(frame_size 2048, fps 20, origin 'future') -> |STFT| with a Hann window -> LogarithmicFilterbank
// (16 bands/octave, 30..6000 Hz: 92 triangular filters) -> log10(1 + x).  One workgroup per frame: windowed frame
// and a 2048-entry twiddle table in LDS, direct DFT of the bins the filterbank touches (<= 558 of 1024: 2.3 MFLOP
// per frame - an FFT would not pay), magnitudes in LDS, filterbank + logarithm.
namespace asr {

struct SpecArgs {
    const float *samples; int64_t n_samples;
    const float *window;            // [frame_size]
    int frame_size; double hop;
    int max_bin;                    // DFT bins 0 .. max_bin-1 are needed
    const int32_t *fb_start, *fb_len, *fb_off;   // per filter: first bin, number of bins, offset into fb_w
    const float *fb_w; int nf;
    float mul, add;
    float *out; int64_t n_frames; int transposed;      // out: (n_frames, nf) or (nf, n_frames)
};

__global__ __launch_bounds__(256) void spectrogram_kernel(SpecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int F = a.frame_size;
    float *x = sl, *tc = sl + F, *ts = sl + 2 * F, *mag = sl + 3 * F;
    const int tid = threadIdx.x;
    for (int t = tid; t < F; t += 256) {
        double s, c;
        sincospi(2.0 * (double)t / (double)F, &s, &c);
        tc[t] = (float)c; ts[t] = (float)s;
    }
    for (int64_t frame = blockIdx.x; frame < a.n_frames; frame += gridDim.x) {
        const int64_t start = (int64_t)((double)frame * a.hop);        // int(index * hop_size), origin 'future'
        __syncthreads();
        for (int t = tid; t < F; t += 256) {
            const int64_t i = start + t;
            x[t] = (i < a.n_samples ? a.samples[i] : 0.0f) * a.window[t];
        }
        __syncthreads();
        for (int k = tid; k < a.max_bin; k += 256) {
            float re = 0.0f, im = 0.0f;
            int idx = 0;
            for (int n = 0; n < F; ++n) {
                re = fmaf(x[n], tc[idx], re);
                im = fmaf(-x[n], ts[idx], im);
                idx = (idx + k) & (F - 1);                              // (k * n) mod F, F a power of two
            }
            mag[k] = sqrtf(re * re + im * im);
        }
        __syncthreads();
        for (int f = tid; f < a.nf; f += 256) {
            const float *w = a.fb_w + a.fb_off[f];
            const int b0 = a.fb_start[f];
            float s = 0.0f;
            for (int q = 0; q < a.fb_len[f]; ++q) s = fmaf(mag[b0 + q], w[q], s);
            const float v = log10f(a.mul * s + a.add);
            if (a.transposed) a.out[(int64_t)f * a.n_frames + frame] = v;
            else a.out[frame * a.nf + f] = v;
        }
    }
}

hipError_t launch_spectrogram(hipStream_t s, const float *samples, int64_t n_samples, const float *window, int frame_size,
                              double hop, int max_bin, const int32_t *fb_start, const int32_t *fb_len,
                              const int32_t *fb_off, const float *fb_w, int nf, float mul, float add, float *out,
                              int64_t n_frames, int transposed) {
    if (n_frames == 0) return hipSuccess;
    SpecArgs a{samples, n_samples, window, frame_size, hop, max_bin, fb_start, fb_len, fb_off, fb_w, nf, mul, add, out,
               n_frames, transposed};
    const size_t lds = (size_t)(3 * frame_size + max_bin) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(spectrogram_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int grid = (int)std::min<int64_t>(n_frames, 2048);
    hipLaunchKernelGGL(spectrogram_kernel, dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

}  // namespace asr

// ---- self-check helpers of the autotuner (ASR_TUNE_VERIFY=1) ---------------------------------------------------
namespace asr {

__global__ __launch_bounds__(256) void fill_pattern_kernel(float *__restrict__ p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t h = (uint32_t)i * 2654435761u;
        p[i] = (float)((h >> 8) & 0xFFFF) * (1.0f / 65536.0f) - 0.25f;      // in [-0.25, 0.75): ELU-like range
    }
}

// *out (uint32 bit pattern of a non-negative float) = max |a - b|; NaN anywhere -> +inf
__global__ __launch_bounds__(256) void max_abs_diff_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           int64_t n, uint32_t *__restrict__ out) {
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d == d) ? fmaxf(m, d) : INFINITY;
    }
    atomicMax(out, __float_as_uint(m));
}

hipError_t launch_fill_pattern(hipStream_t s, float *p, int64_t n) {
    if (n <= 0) return hipSuccess;
    fill_pattern_kernel<<<(int)std::min<int64_t>((n + 255) / 256, 4096), 256, 0, s>>>(p, n);
    return hipGetLastError();
}

hipError_t launch_max_abs_diff(hipStream_t s, const float *a, const float *b, int64_t n, uint32_t *out_bits) {
    hipError_t e = hipMemsetAsync(out_bits, 0, sizeof(uint32_t), s);
    if (e != hipSuccess || n <= 0) return e;
    max_abs_diff_kernel<<<(int)std::min<int64_t>((n + 255) / 256, 4096), 256, 0, s>>>(a, b, n, out_bits);
    return hipGetLastError();
}

}  // namespace asr
