// K1 -- monotonic alignment search on gfx950 (CDNA4), replacing
//   S_monotonic_align.py:5-47   (maximum_path1: tie -> move up)
//   S_monotonic_align.py:50-95  (maximum_path2: tie -> stay)
//   S_monotonic_align_Triton.py:7-71 (Triton kernel, tie -> stay)   utils.py:11-24 (Cython wrapper)
//
// One workgroup per utterance.  Thread t owns R consecutive lattice rows [t*R, t*R+R); the previous
// DP column lives in registers, the row above a lane's first row comes from the neighbouring lane
// (wave shuffle) or, across waves, through a 2-slot LDS mailbox.  Each thread streams ITS rows along
// the contiguous mel axis 16 bytes at a time (one dwordx4 per row per 4 columns, next chunk
// prefetched while the current one is consumed), so every fetched line is fully used -- the
// Triton kernel instead reads a stride-Ty column per step.
//
// The DP does exactly one fp32 add per cell, v = value + max(a, c), with -1e32 sentinels, i.e. the
// reference's arithmetic; decisions are bit-exact with it (NaN inputs excepted).  Instead of storing
// the cumulative lattice (the reference's in-place 4 B/cell) we keep 1 bit per cell ("came from the
// row above"): each lane shifts its R decision bits per column into a register and stores one 64-bit
// word per 4 columns (coalesced, 512 B per wave).  Per cell that is 4-5 vector instructions: compare,
// select, add, shift-in.  The backtrack stages those words through LDS in column chunks and one lane
// walks them; the dense 0/1 path, the per-column row index and the integer durations are emitted.
//
// Algorithmic bytes: 4*Tx*Ty read per utterance + 4*Tx*Ty path written (memset) +
// Tx*Ty/8 decision bits written and read.
#include "common.h"
#include "artspeech_hip.h"

// the DP must do exactly one fp32 add per cell: no fused multiply-add anywhere in this file
#pragma clang fp contract(off)

#define MAS_NEG (-1e32f)
typedef unsigned long long u64;

template <int R, bool VEC4, bool TIE_MOVE, int MAXT>
__global__ void __launch_bounds__(MAXT)
mas_kernel(const float* __restrict__ value, const int* __restrict__ t_x, const int* __restrict__ t_y,
           int Tx, int Ty, float* __restrict__ path, int* __restrict__ dur, int* __restrict__ rows,
           u64* __restrict__ ws, int stage_chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int NT = blockDim.x;
    const int W = NT >> 6;
    u64* stage = reinterpret_cast<u64*>(smem_raw);                       // [stage_chunks][NT]
    float* bnd = reinterpret_cast<float*>(stage + (size_t)stage_chunks * NT);   // [2][16]
    int* rowbuf = reinterpret_cast<int*>(bnd + 2 * 16);                  // [4*stage_chunks]

    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    if (x_len <= 0 || y_len <= 0) return;                  // uniform per block

    const float* vb = value + (size_t)b * Tx * Ty;
    const int nchunk_max = (Ty + 3) >> 2;
    u64* wsb = ws + (size_t)b * nchunk_max * NT;
    const int r0 = tid * R;

    float prev[R];
    float4 cur[R], nxt[R];

    auto load_chunk = [&](int c, float4 (&dst)[R]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int x = r0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < x_len) {
                const float* p = vb + (size_t)x * Ty + 4 * c;
                if (VEC4) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    const int rem = Ty - 4 * c;
                    v.x = p[0];
                    if (rem > 1) v.y = p[1];
                    if (rem > 2) v.z = p[2];
                    if (rem > 3) v.w = p[3];
                }
            }
            dst[r] = v;
        }
    };

    const int nchunk = (y_len + 3) >> 2;
    load_chunk(0, cur);
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) load_chunk(c + 1, nxt);
        unsigned bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = 4 * c + j;
            if (y >= y_len) break;
            if (y == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) prev[r] = (r0 + r == 0) ? cur[0].x : MAS_NEG;
            } else {
                float up = __shfl_up(prev[R - 1], 1);             // (a DPP wave_shr:1 instead of this ds_bpermute measures the same: 808 us)
                if (lane == 0) up = (wave == 0) ? MAS_NEG : bnd[((y - 1) & 1) * 16 + wave - 1];
                float nv[R];
                unsigned bj = 0u;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float a = prev[r];
                    const float cc = (r > 0) ? prev[r - 1] : up;
                    // v1: direction = where(a > c, 0, -1);  v2/Triton: move iff c > a.  max = where(a > c, a, c).
                    const bool move = TIE_MOVE ? !(a > cc) : (cc > a);
                    const float m = move ? cc : a;
                    bj = (bj << 1) | (move ? 1u : 0u);         // row r ends at bit R-1-r
                    const float val = (j == 0) ? cur[r].x : (j == 1) ? cur[r].y : (j == 2) ? cur[r].z : cur[r].w;
                    nv[r] = __fadd_rn(val, m);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) prev[r] = nv[r];
                bits[j] = bj;
            }
            if (W > 1) {
                if (lane == 63) bnd[(y & 1) * 16 + wave] = prev[R - 1];
                // LDS-only barrier: __syncthreads() would also wait for the prefetched chunk's global loads (vmcnt)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
        wsb[(size_t)c * NT + tid] = (u64)(bits[0] | (bits[1] << 16)) | ((u64)(bits[2] | (bits[3] << 16)) << 32);
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
    }

    // ---- backtrack ------------------------------------------------------------------------------
    __syncthreads();                                       // decision words visible to the block
    int idx = x_len - 1;
    int run = 1;                                           // cells of the path in row idx so far
    if (tid == 0) {
        if (path) path[((size_t)b * Tx + idx) * Ty + (y_len - 1)] = 1.f;
        if (rows) rows[(size_t)b * Ty + (y_len - 1)] = idx;
    }
    for (int c_hi = (y_len - 1) >> 2; c_hi >= 0; c_hi -= stage_chunks) {
        const int c_lo = (c_hi - stage_chunks + 1) > 0 ? (c_hi - stage_chunks + 1) : 0;
        const int nwords = (c_hi - c_lo + 1) * NT;
        for (int i = tid; i < nwords; i += NT) stage[i] = wsb[(size_t)c_lo * NT + i];
        __syncthreads();
        const int y_top = (4 * c_hi + 3) < (y_len - 1) ? (4 * c_hi + 3) : (y_len - 1);
        const int y_bot = (4 * c_lo) > 1 ? (4 * c_lo) : 1;
        if (tid == 0) {
            for (int y = y_top; y >= y_bot; --y) {
                const int t = idx / R, r = idx - t * R;
                const u64 w = stage[((y >> 2) - c_lo) * NT + t];
                const int move = (int)((w >> (16 * (y & 3) + (R - 1 - r))) & 1ull);
                if (move && idx > 0) {
                    if (dur) dur[(size_t)b * Tx + idx] = run;
                    idx -= 1;
                    run = 0;
                }
                run += 1;
                rowbuf[y - y_bot] = idx;                   // row of column y-1
            }
        }
        __syncthreads();
        for (int i = tid; i <= y_top - y_bot; i += NT) {
            const int col = y_bot + i - 1;
            const int rr = rowbuf[i];
            if (path) path[((size_t)b * Tx + rr) * Ty + col] = 1.f;
            if (rows) rows[(size_t)b * Ty + col] = rr;
        }
        __syncthreads();
    }
    if (tid == 0 && dur) dur[(size_t)b * Tx + idx] = run;
}

// ---- host side -------------------------------------------------------------------------------------
static int mas_geometry(int Tx, int* R, int* W)
{
    // rows per lane R and waves W (64 * R * W >= Tx).  A column step costs ~5 instructions per row of the lane plus,
    // with several waves, one LDS-only barrier: measured on MI355X ([8,1024,2000]) 1 wave x 16 rows 1.24 ms, 8 x 2
    // 0.80 ms -- so as many waves as the workgroup allows (16 at <= 128 VGPRs for R <= 4, 8 for R = 8 / 16).
    int r = 1;
    const char* env = getenv("AS_MAS_R");                  // tuning/experiments only
    if (env && atoi(env) > 0) {
        r = atoi(env);
        if (r != 1 && r != 2 && r != 4 && r != 8 && r != 16) return AS_EINVAL;
    } else {
        // eight waves at most unless the rows do not fit otherwise: [8,1024,2000] 8 waves x 2 rows 0.80 ms, 16 x 1 0.91, 4 x 4 0.86,
        // 2 x 8 0.95, 1 x 16 1.25 (a variant without the per-column barrier -- skewed waves handing their boundary row over through an
        // LDS ring -- was 0.1 ms SLOWER at every geometry: the column time is the waves' own dependent chains, not the barrier)
        while (r < 16 && 64 * r * 8 < Tx) r <<= 1;
    }
    const int w = as_cdiv(Tx > 0 ? Tx : 1, 64 * r);
    if (w > (r <= 4 ? 16 : 8)) return AS_EINVAL;           // Tx <= 8192
    *R = r;
    *W = w;
    return AS_OK;
}

extern "C" size_t as_mas_workspace_bytes(int B, int Tx, int Ty)
{
    int R, W;
    if (B <= 0 || Tx <= 0 || Ty <= 0 || mas_geometry(Tx, &R, &W) != AS_OK) return 0;
    return (size_t)B * ((Ty + 3) / 4) * 64 * W * sizeof(u64);
}

template <int R, bool TIE>
static void mas_launch(bool vec4, int B, int W, size_t smem, hipStream_t s, const float* value, const int* t_x,
                       const int* t_y, int Tx, int Ty, float* path, int* dur, int* rows, u64* ws, int sc)
{
    // W == 1 (the common case, Tx <= 64*R): 64-thread workgroups may use the whole register file, which
    // the R = 16 prefetch needs; multi-wave geometries go up to 16 waves (128 VGPRs each).
    if (W == 1) {
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, true, TIE, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
        else
            hipLaunchKernelGGL((mas_kernel<R, false, TIE, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
    } else {
        constexpr int MT = R <= 4 ? 1024 : 512;            // launch bound = register budget: R = 8 / 16 need 256 VGPRs
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, true, TIE, MT>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
        else
            hipLaunchKernelGGL((mas_kernel<R, false, TIE, MT>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
    }
}

template <bool TIE>
static void mas_dispatch(int R, bool vec4, int B, int W, size_t smem, hipStream_t s, const float* value, const int* t_x,
                         const int* t_y, int Tx, int Ty, float* path, int* dur, int* rows, u64* ws, int sc)
{
    switch (R) {
    case 1: mas_launch<1, TIE>(vec4, B, W, smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc); break;
    case 2: mas_launch<2, TIE>(vec4, B, W, smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc); break;
    case 4: mas_launch<4, TIE>(vec4, B, W, smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc); break;
    case 8: mas_launch<8, TIE>(vec4, B, W, smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc); break;
    default: mas_launch<16, TIE>(vec4, B, W, smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc); break;
    }
}

extern "C" int as_mas_f32(const float* value, const int* t_x, const int* t_y, int B, int Tx, int Ty,
                          int tie_mode, float* path, int* dur, int* rows, void* ws, size_t ws_bytes,
                          as_stream_t stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!value || !t_x || !t_y || B < 0 || Tx <= 0 || Ty <= 0 || (tie_mode != 0 && tie_mode != 1)) return AS_EINVAL;
    if (B == 0) return AS_OK;
    int R, W;
    if (mas_geometry(Tx, &R, &W) != AS_OK) return AS_EINVAL;
    const size_t need = (size_t)B * ((Ty + 3) / 4) * 64 * W * sizeof(u64);
    if (!ws || ws_bytes < need) return AS_EINVAL;
    if (path) AS_CHECK(hipMemsetAsync(path, 0, (size_t)B * Tx * Ty * sizeof(float), stream));
    if (dur) AS_CHECK(hipMemsetAsync(dur, 0, (size_t)B * Tx * sizeof(int), stream));
    if (rows) AS_CHECK(hipMemsetAsync(rows, 0xFF, (size_t)B * Ty * sizeof(int), stream));
    const int NT = 64 * W;
    int sc = 4096 / NT;                                    // <= 32 KiB of staged decision words
    sc = sc > 16 ? 16 : sc;
    const size_t smem = (size_t)sc * NT * 8 + 2 * 16 * sizeof(float) + (size_t)4 * sc * sizeof(int);
    const bool vec4 = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0);
    u64* w64 = static_cast<u64*>(ws);
    AsProfScope prof__(AS_CLS_MAS, 2.0 * B * Tx * (double)Ty, 4.0 * B * Tx * (double)Ty * (path ? 2 : 1), stream);
    if (tie_mode) mas_dispatch<true>(R, vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, path, dur, rows, w64, sc);
    else mas_dispatch<false>(R, vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, path, dur, rows, w64, sc);
    AS_CHECK_LAUNCH();
    return AS_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// SURVEY.md section 8(f) N4 -- the training scripts' producer of K1 (train_second.py:181-184, train_first.py:171-177):
//   s2s_attn = softmax(s2s_attn_feat, dim = -1 | 1);  mask_ST = mask_from_lens(...);  mono = maximum_path(s2s_attn, mask_ST);
//   d_gt = mono.sum(-1)
// as one entry point that takes the LENGTHS (no dense mask is built): a softmax kernel over the whole [Tx][Ty] slab of every
// item (as the reference: no masking before the softmax in train_second.py), then the MAS kernels above on it.
// ----------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
softmax_last_kernel(const float* __restrict__ x, int Ty, float* __restrict__ y)
{
    __shared__ float red[4];
    const float* xr = x + (size_t)blockIdx.x * Ty;
    float* yr = y + (size_t)blockIdx.x * Ty;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < Ty; i += 256) m = fmaxf(m, xr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < Ty; i += 256) s += expf(xr[i] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = threadIdx.x; i < Ty; i += 256) yr[i] = expf(xr[i] - m) / s;
}

// softmax over dim 1 (the Tx axis): one thread per (item, column), two passes over the column (coalesced across threads)
__global__ void softmax_dim1_kernel(const float* __restrict__ x, int Tx, int Ty, float* __restrict__ y)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= Ty) return;
    const float* xb = x + (size_t)b * Tx * Ty + j;
    float* yb = y + (size_t)b * Tx * Ty + j;
    float m = -INFINITY;
    for (int i = 0; i < Tx; ++i) m = fmaxf(m, xb[(size_t)i * Ty]);
    float s = 0.f;
    for (int i = 0; i < Tx; ++i) s += expf(xb[(size_t)i * Ty] - m);
    for (int i = 0; i < Tx; ++i) yb[(size_t)i * Ty] = expf(xb[(size_t)i * Ty] - m) / s;
}

extern "C" int as_softmax_mas_f32(const float* feat, const int* t_x, const int* t_y, int B, int Tx, int Ty, int softmax_dim,
                                  int tie_mode, float* attn, float* path, int* dur, int* rows, void* ws, size_t ws_bytes,
                                  as_stream_t stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!feat || !attn || (softmax_dim != 1 && softmax_dim != 2) || B < 0 || Tx <= 0 || Ty <= 0) return AS_EINVAL;
    if (B == 0) return AS_OK;
    {
        AsProfScope prof__(AS_CLS_MAS, 0, 8.0 * B * (double)Tx * Ty, stream);
        if (softmax_dim == 2) hipLaunchKernelGGL(softmax_last_kernel, dim3(B * Tx), dim3(256), 0, stream, feat, Ty, attn);
        else hipLaunchKernelGGL(softmax_dim1_kernel, dim3(as_cdiv(Ty, 256), B), dim3(256), 0, stream, feat, Tx, Ty, attn);
        AS_CHECK_LAUNCH();
    }
    return as_mas_f32(attn, t_x, t_y, B, Tx, Ty, tie_mode, path, dur, rows, ws, ws_bytes, stream_);
}
