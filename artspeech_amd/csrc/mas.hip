// K1 -- monotonic alignment search on gfx950 (CDNA4), replacing
//   S_monotonic_align.py:5-47   (maximum_path1: tie -> move up)
//   S_monotonic_align.py:50-95  (maximum_path2: tie -> stay)
//   S_monotonic_align_Triton.py:7-71 (Triton kernel, tie -> stay)   utils.py:11-24 (Cython wrapper)
//
// One workgroup per utterance.  Thread t owns R consecutive lattice rows [t*R, t*R+R); the previous
// DP column lives in registers, the row above a lane's first row comes from the neighbouring lane
// (wave shuffle) or, across waves, through a 2-slot LDS mailbox.  Each thread streams ITS rows along
// the contiguous mel axis 16 bytes at a time (one dwordx4 per row per 4 columns, next chunk
// prefetched while the current one is consumed), so every fetched 64-B line is fully used -- the
// Triton kernel instead reads a stride-Ty column per step.
//
// The DP does exactly one fp32 add per cell, v = value + (a > c ? a : c), with -1e32 sentinels,
// i.e. the reference's arithmetic; decisions are bit-exact with it.  Instead of storing the
// cumulative lattice (the reference's in-place 4 B/cell) we keep 1 bit per cell: the compare that
// feeds the max IS the wave-wide ballot (v_cmp writes the 64-lane mask), stored as R 64-bit words
// per wave per column.  The backtrack stages those words through LDS in column chunks and one lane
// walks them; the dense 0/1 path, the per-column row index and the integer durations are emitted.
//
// Algorithmic bytes: 4*Tx*Ty read per utterance + 4*Tx*Ty path written (memset) +
// Tx*Ty/8 decision bits written and read.
#include "common.h"
#include "artspeech_hip.h"

#define MAS_NEG (-1e32f)

template <int R, bool VEC4, int MAXT>
__global__ void __launch_bounds__(MAXT)
mas_kernel(const float* __restrict__ value, const int* __restrict__ t_x, const int* __restrict__ t_y,
           int Tx, int Ty, int tie_move, float* __restrict__ path, int* __restrict__ dur,
           int* __restrict__ rows, unsigned long long* __restrict__ ws, int chunk_cols)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int W = blockDim.x >> 6;
    const int nw = W * R;                                  // decision words per column
    unsigned long long* stage = reinterpret_cast<unsigned long long*>(smem_raw);
    float* bnd = reinterpret_cast<float*>(stage + (size_t)chunk_cols * nw);   // [2][W]
    int* rowbuf = reinterpret_cast<int*>(bnd + 2 * 16);                      // [chunk_cols]

    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    if (x_len <= 0 || y_len <= 0) return;                  // uniform per block

    const float* vb = value + (size_t)b * Tx * Ty;
    unsigned long long* wsb = ws + (size_t)b * Ty * nw;
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
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = 4 * c + j;
            if (y >= y_len) break;
            if (y == 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) prev[r] = (r0 + r == 0) ? cur[0].x : MAS_NEG;
            } else {
                float up = __shfl_up(prev[R - 1], 1);
                if (lane == 0) up = (wave == 0) ? MAS_NEG : bnd[((y - 1) & 1) * 16 + wave - 1];
                unsigned long long mine = 0ull;
                float nv[R];
#pragma unroll
                for (int r = R - 1; r >= 0; --r) {
                    const float a = prev[r];
                    const float cc = (r > 0) ? prev[r - 1] : up;
                    const bool gt = a > cc;
                    const float m = gt ? a : cc;
                    const bool move = tie_move ? !gt : (cc > a);
                    const unsigned long long bal = __ballot(move);
                    if (lane == r) mine = bal;
                    const float val = (j == 0) ? cur[r].x : (j == 1) ? cur[r].y : (j == 2) ? cur[r].z : cur[r].w;
                    nv[r] = __fadd_rn(val, m);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) prev[r] = nv[r];
                if (lane < R) wsb[(size_t)y * nw + wave * R + lane] = mine;
            }
            if (W > 1) {
                if (lane == 63) bnd[(y & 1) * 16 + wave] = prev[R - 1];
                __syncthreads();
            }
        }
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
    for (int y_hi = y_len - 1; y_hi >= 1; y_hi -= chunk_cols) {
        const int y_lo = (y_hi - chunk_cols + 1) > 1 ? (y_hi - chunk_cols + 1) : 1;
        const int ncol = y_hi - y_lo + 1;
        for (int i = tid; i < ncol * nw; i += blockDim.x) stage[i] = wsb[(size_t)y_lo * nw + i];
        __syncthreads();
        if (tid == 0) {
            for (int y = y_hi; y >= y_lo; --y) {
                const int t = idx / R, r = idx - t * R;
                const unsigned long long w = stage[(size_t)(y - y_lo) * nw + (t >> 6) * R + r];
                const int move = (int)((w >> (t & 63)) & 1ull);
                if (move && idx > 0) {
                    if (dur) dur[(size_t)b * Tx + idx] = run;
                    idx -= 1;
                    run = 0;
                }
                run += 1;
                rowbuf[y - y_lo] = idx;                    // row of column y-1
            }
        }
        __syncthreads();
        for (int i = tid; i < ncol; i += blockDim.x) {
            const int col = y_lo + i - 1;
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
    int r = 1;
    const char* env = getenv("AS_MAS_R");                  // tuning/experiments only
    if (env && atoi(env) > 0) {
        r = atoi(env);
        if (r != 1 && r != 2 && r != 4 && r != 8 && r != 16) return AS_EINVAL;
    } else {
        while (r < 16 && 64 * r < Tx) r <<= 1;
    }
    const int w = as_cdiv(Tx > 0 ? Tx : 1, 64 * r);
    if (w > 8) return AS_EINVAL;                           // Tx <= 8192 at R = 16
    *R = r;
    *W = w;
    return AS_OK;
}

static int mas_chunk_cols(int nw)
{
    int ch = 4096 / nw;                                    // <= 32 KiB of staged decision words
    return ch > 64 ? 64 : (ch < 1 ? 1 : ch);
}

extern "C" size_t as_mas_workspace_bytes(int B, int Tx, int Ty)
{
    int R, W;
    if (B <= 0 || Tx <= 0 || Ty <= 0 || mas_geometry(Tx, &R, &W) != AS_OK) return 0;
    return (size_t)B * Ty * W * R * sizeof(unsigned long long);
}

template <int R>
static void mas_launch(bool vec4, int B, int W, size_t smem, hipStream_t s, const float* value, const int* t_x,
                       const int* t_y, int Tx, int Ty, int tie, float* path, int* dur, int* rows,
                       unsigned long long* ws, int ch)
{
    // W == 1 (the common case, Tx <= 64*R): 64-thread workgroups may use the whole register file, which
    // the R = 16 prefetch needs; multi-wave geometries are capped at 8 waves (256 VGPRs each).
    if (W == 1) {
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, true, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, tie, path, dur, rows, ws, ch);
        else
            hipLaunchKernelGGL((mas_kernel<R, false, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, tie, path, dur, rows, ws, ch);
    } else {
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, true, 512>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, tie, path, dur, rows, ws, ch);
        else
            hipLaunchKernelGGL((mas_kernel<R, false, 512>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, tie, path, dur, rows, ws, ch);
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
    const size_t need = (size_t)B * Ty * W * R * sizeof(unsigned long long);
    if (!ws || ws_bytes < need) return AS_EINVAL;
    if (path) AS_CHECK(hipMemsetAsync(path, 0, (size_t)B * Tx * Ty * sizeof(float), stream));
    if (dur) AS_CHECK(hipMemsetAsync(dur, 0, (size_t)B * Tx * sizeof(int), stream));
    if (rows) AS_CHECK(hipMemsetAsync(rows, 0xFF, (size_t)B * Ty * sizeof(int), stream));
    const int ch = mas_chunk_cols(W * R);
    const size_t smem = (size_t)ch * W * R * 8 + 2 * 16 * sizeof(float) + (size_t)ch * sizeof(int);
    const bool vec4 = (Ty % 4 == 0) && ((reinterpret_cast<uintptr_t>(value) & 15) == 0);
    unsigned long long* w64 = static_cast<unsigned long long*>(ws);
    AsProfScope prof__(AS_CLS_MAS, 2.0 * B * Tx * (double)Ty, 4.0 * B * Tx * (double)Ty * (path ? 2 : 1), stream);
    switch (R) {
    case 1: mas_launch<1>(vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, tie_mode, path, dur, rows, w64, ch); break;
    case 2: mas_launch<2>(vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, tie_mode, path, dur, rows, w64, ch); break;
    case 4: mas_launch<4>(vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, tie_mode, path, dur, rows, w64, ch); break;
    case 8: mas_launch<8>(vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, tie_mode, path, dur, rows, w64, ch); break;
    default: mas_launch<16>(vec4, B, W, smem, stream, value, t_x, t_y, Tx, Ty, tie_mode, path, dur, rows, w64, ch); break;
    }
    AS_CHECK_LAUNCH();
    return AS_OK;
}
