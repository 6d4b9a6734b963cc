// Probe (run ON THE GPU BOX: hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_bf16x3_probe.hip && /tmp/probe):
// the cosine of unit-length 32-d rows on the bf16 MFMA as a THREE-PLANE split with six cross terms, against the native
// fp32 MFMA (eight v_mfma_f32_16x16x4_f32) and a float64 reference - the error each form makes and the time per tile.
//   x = x1 + x2 + x3 exactly (x1 = top 8 significant bits of the fp32 value by truncation, x2 the next 8, x3 the rest),
//   <x, y> ~ x3.y1 + x1.y3 + x2.y2 + x2.y1 + x1.y2 + x1.y1  (dropped: x2.y3, x3.y2, x3.y3 <= 2^-25 |x||y|)
// one v_mfma_f32_16x16x32_bf16 per term (K = 32 = the whole row).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const float (&x)[8], bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
    unsigned u1[4], u2[4], u3[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        const unsigned a1 = __float_as_uint(a) & 0xFFFF0000u, b1 = __float_as_uint(b) & 0xFFFF0000u;
        const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
        const unsigned a2 = __float_as_uint(ra) & 0xFFFF0000u, b2 = __float_as_uint(rb) & 0xFFFF0000u;
        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
        u1[i] = (a1 >> 16) | b1;
        u2[i] = (a2 >> 16) | b2;
        u3[i] = (__float_as_uint(sa) >> 16) | (__float_as_uint(sb) & 0xFFFF0000u);
    }
    const uintx4 v1 = {u1[0], u1[1], u1[2], u1[3]}, v2 = {u2[0], u2[1], u2[2], u2[3]}, v3 = {u3[0], u3[1], u3[2], u3[3]};
    p1 = __builtin_bit_cast(bf16x8, v1);
    p2 = __builtin_bit_cast(bf16x8, v2);
    p3 = __builtin_bit_cast(bf16x8, v3);
}

// one wave per tile pair: items [16][32], queries [16][32] -> out[16 items][16 queries]; lane (g = lane >> 4, m = lane & 15)
// holds dims 8g .. 8g+7 of item m (A) and of query m (B); C: lane (g, n) holds items 4g + r against query n
__global__ void probe_kernel(const float *items, const float *queries, float *out3, float *out32, int tiles, int reps,
                             int mode) {
    const int lane = threadIdx.x & 63, g = lane >> 4, m = lane & 15;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= tiles) return;
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = items[((size_t)t * 16 + m) * 32 + 8 * g + j];
        b[j] = queries[((size_t)t * 16 + m) * 32 + 8 * g + j];
    }
    bf16x8 b1, b2, b3;
    split3(b, b1, b2, b3);
    floatx4 acc3 = {0.f, 0.f, 0.f, 0.f}, acc32 = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        if (mode & 1) {
            bf16x8 a1, a2, a3;
            split3(a, a1, a2, a3);
            floatx4 c = {0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b3, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b2, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c, 0, 0, 0);
            acc3 = c;
        }
        if (mode & 2) {
            floatx4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
            acc32 = c;
        }
        if (reps > 1) {                       // keep the loop from being hoisted: the operands depend on the results
            a[0] += 1e-30f * (acc3[0] + acc32[0]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        out3[((size_t)t * 16 + 4 * g + r) * 16 + m] = acc3[r];
        out32[((size_t)t * 16 + 4 * g + r) * 16 + m] = acc32[r];
    }
}

static void unit(float *v) {
    double n = 0;
    for (int j = 0; j < 32; ++j) n += (double)v[j] * v[j];
    const float r = (float)(1.0 / std::sqrt(n));
    for (int j = 0; j < 32; ++j) v[j] *= r;
}

int main() {
    const int tiles = 1 << 16;
    std::vector<float> it((size_t)tiles * 512), qu((size_t)tiles * 512);
    srand(7);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int t = 0; t < tiles; ++t)
        for (int m = 0; m < 16; ++m) {
            float *a = &it[((size_t)t * 16 + m) * 32], *b = &qu[((size_t)t * 16 + m) * 32];
            const int kind = t & 7;
            for (int j = 0; j < 32; ++j) {
                if (kind == 0) { a[j] = 1.f; b[j] = 1.f + 1e-3f * rnd(); }                  // all of one sign: sum |ab| = 1
                else if (kind == 1) { a[j] = (j & 1) ? -1.f : 1.f; b[j] = 1.f + 0.3f * rnd(); }  // cancellation
                else if (kind == 2) { a[j] = rnd(); b[j] = a[j] + 1e-2f * rnd(); }          // near neighbours
                else if (kind == 3) { a[j] = j == (m & 31) ? 1.f : 1e-4f * rnd(); b[j] = j == (m & 31) ? 1.f : 1e-4f * rnd(); }
                else { a[j] = rnd(); b[j] = rnd(); }
            }
            unit(a); unit(b);
        }
    float *di, *dq, *d3, *d32;
    hipMalloc(&di, it.size() * 4); hipMalloc(&dq, qu.size() * 4);
    hipMalloc(&d3, (size_t)tiles * 256 * 4); hipMalloc(&d32, (size_t)tiles * 256 * 4);
    hipMemcpy(di, it.data(), it.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, qu.data(), qu.size() * 4, hipMemcpyHostToDevice);
    probe_kernel<<<tiles / 4, 256>>>(di, dq, d3, d32, tiles, 1, 3);
    std::vector<float> o3((size_t)tiles * 256), o32((size_t)tiles * 256);
    hipMemcpy(o3.data(), d3, o3.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(o32.data(), d32, o32.size() * 4, hipMemcpyDeviceToHost);
    double e3[8] = {0}, e32[8] = {0};
    for (int t = 0; t < tiles; ++t)
        for (int i = 0; i < 16; ++i)
            for (int q = 0; q < 16; ++q) {
                double ref = 0;
                for (int j = 0; j < 32; ++j) ref += (double)it[((size_t)t * 16 + i) * 32 + j] * qu[((size_t)t * 16 + q) * 32 + j];
                const double a = std::fabs(o3[((size_t)t * 16 + i) * 16 + q] - ref), b = std::fabs(o32[((size_t)t * 16 + i) * 16 + q] - ref);
                if (a > e3[t & 7]) e3[t & 7] = a;
                if (b > e32[t & 7]) e32[t & 7] = b;
            }
    const char *names[8] = {"one sign", "alternating", "near neighbours", "one-hot", "random", "random", "random", "random"};
    for (int k = 0; k < 5; ++k)
        printf("max |dot - float64|, %-16s bf16 x 3 planes (6 terms): %.3e    fp32 MFMA: %.3e\n", names[k], e3[k], e32[k]);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 1; mode <= 2; ++mode) {
        const int reps = 2000;
        probe_kernel<<<1024, 256>>>(di, dq, d3, d32, 4096, reps, mode);
        hipEventRecord(e0);
        probe_kernel<<<1024, 256>>>(di, dq, d3, d32, 4096, reps, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        // 1024 workgroups x 4 waves on 256 CUs x 4 SIMDs: 4 waves per SIMD, each `reps` tiles
        printf("%s: %.3f ms for %d tiles per wave, 4 waves per SIMD -> %.1f ns per tile and SIMD (incl. the split of A per tile)\n",
               mode == 1 ? "bf16 x 3 planes" : "fp32 MFMA      ", ms, reps, ms * 1e6 / (reps * 4.0));
    }
    return 0;
}
