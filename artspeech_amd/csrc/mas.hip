// K1 -- monotonic alignment search on gfx950 (CDNA4), replacing
//   S_monotonic_align.py:5-47   (maximum_path1: tie -> move up)
//   S_monotonic_align.py:50-95  (maximum_path2: tie -> stay)
//   S_monotonic_align_Triton.py:7-71 (Triton kernel, tie -> stay)   utils.py:11-24 (Cython wrapper)
//
// Two kernels.  Rows that are 16-byte aligned (Ty % 4 == 0: the usual case) take the BANDED kernel further down (a workgroup per band
// of 128 rows, DP waves chained down the rows, LDS-DMA loader wave, wave-parallel backtrack: [8,1024,2000] in 0.14 ms).  Everything
// else takes the round-1 kernel described here (0.80 ms on the same lattice):
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
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "artspeech_hip.h"

// the DP must do exactly one fp32 add per cell: no fused multiply-add anywhere in this file
#pragma clang fp contract(off)

#define MAS_NEG (-1e32f)
typedef unsigned long long u64;

#ifdef AS_EXPERIMENTS
__device__ unsigned long long mas_dbg[8];      // shader / wall clocks at the phase boundaries of block 0 (scripts/exp/mas_clock.py)
#define MAS_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) { mas_dbg[2 * (i)] = clock64(); mas_dbg[2 * (i) + 1] = wall_clock64(); } } while (0)
__device__ unsigned long long mas_dbg2[128];    // wall clock (100 MHz) per (band, wave) of utterance 0: block 0 may start / is done, last block done
extern "C" int as_mas_debug(unsigned long long* out8) { return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(mas_dbg), 64); }
extern "C" int as_mas_debug2(unsigned long long* out128) { return (int)hipMemcpyFromSymbol(out128, HIP_SYMBOL(mas_dbg2), 1024); }
#else
#define MAS_MARK(i)
#endif

template <int R, int Q, bool VEC4, bool TIE_MOVE, int MAXT>
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
    const int ngroup_max = (Ty + 3) >> 2;                  // a "group" = 4 columns = one decision word per thread
    u64* wsb = ws + (size_t)b * ngroup_max * NT;
    const int r0 = tid * R;

    float prev[R];
    float4 cur[R][Q], nxt[R][Q];

    // a chunk = Q groups = 16 Q bytes of each of the lane's rows.  Q = 8 is one whole 128-byte line per row: the workgroup's rows are
    // Tx lines 4 Ty bytes apart (128 KB of lines at Tx = 1024, four times the CU's L1), so a line that is fetched 16 bytes at a time
    // comes up from L2 eight times -- [8,1024,2000] took 0.80 ms that way (0.4 us per column, all of it this traffic), whatever the
    // geometry of the waves.
    auto load_chunk = [&](int c, float4 (&dst)[R][Q]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int x = r0 + r;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int g = c * Q + q;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (x < x_len && 4 * g < Ty) {
                    const float* p = vb + (size_t)x * Ty + 4 * g;
                    if (VEC4) {
                        v = *reinterpret_cast<const float4*>(p);
                    } else {
                        const int rem = Ty - 4 * g;
                        v.x = p[0];
                        if (rem > 1) v.y = p[1];
                        if (rem > 2) v.z = p[2];
                        if (rem > 3) v.w = p[3];
                    }
                }
                dst[r][q] = v;
            }
        }
    };

    const int ngroup = (y_len + 3) >> 2;
    const int nchunk = (ngroup + Q - 1) / Q;
    load_chunk(0, cur);
    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) load_chunk(c + 1, nxt);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int g = c * Q + q;
            if (g >= ngroup) break;
            unsigned bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int y = 4 * g + j;
                if (y >= y_len) break;
                if (y == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) prev[r] = (r0 + r == 0) ? cur[0][0].x : MAS_NEG;
                } else {
                    float up = __shfl_up(prev[R - 1], 1);             // (a DPP wave_shr:1 instead of this ds_bpermute measures the same)
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
                        const float val = (j == 0) ? cur[r][q].x : (j == 1) ? cur[r][q].y : (j == 2) ? cur[r][q].z : cur[r][q].w;
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
            wsb[(size_t)g * NT + tid] = (u64)(bits[0] | (bits[1] << 16)) | ((u64)(bits[2] | (bits[3] << 16)) << 32);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < Q; ++q) cur[r][q] = nxt[r][q];
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

// ----------------------------------------------------------------------------------------------------------------
// The banded variant (rows 16-byte aligned): an utterance's lattice is cut into bands of 64 W rows, one workgroup per band, all
// bands of an utterance on one XCD.  A workgroup is W DP waves (64 rows each, one row per lane) and a loader wave; every DP wave runs a
// block or two of columns behind the one above it -- a software pipeline down the rows:
//   the loader:  LDS-DMA (buffer_load ... lds) of the band's rows, three 32-column blocks in flight, into a six-block ring in LDS; an
//                instruction fetches 8 rows x 128 bytes (whole lines).  HBM latency never reaches the DP.
//   a DP wave:   its row of the previous column in a register, the row above through DPP wave_shr:1, no barrier.  A column is ~10
//                instructions: the compare, v_addc (decision word = 2 word + the compare's lane mask), v_max, v_add, the DPP move, one
//                ds_write of the new value (lane 63's is the wave's last row) and one ds_read of the row above the wave.
//   wave -> wave below, same workgroup: the last row's 32 values of a block through LDS, a counter per wave.
//   band -> band below: through 8-byte {value, tag} words in global memory, written once per launch and polled by the lane that needs
//                them (no flag, no fence), requested one block early so that the round trip to L2 is off the critical path.
// Producers have lower workgroup ids than their consumers and never wait for them, so the chain cannot deadlock whatever is resident.
// One workgroup per utterance did 0.4 us per column on [8,1024,2000] whatever its geometry: 1024 rows x 16 bytes per 4 columns is 64
// distinct lines per load instruction, and the texture path of ONE CU took as long over them as the DP itself.
// The backtrack is its own launch (one wave per utterance): the 32 decisions of a block of columns are one word per row; lane l of the
// wave holds the word of row (current row - l), the walk is v_readlane + bit extract + add in scalar registers, and the next block's
// words are in flight meanwhile.
// ----------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void mas_lds_void;
#define MAS_BLK 32            // columns per block
#define MAS_RING 6            // blocks in the LDS ring: ahead + the one in use + the lag of the band's second wave (a block) + slack
#define MAS_AHEAD 3           // blocks of DMA in flight per loader (the ring keeps more slots than ahead + current: the DP never waits for the
                              // loader's handshake)
#define MAS_CHUNK_F 260       // floats per chunk in the ring: 8 rows x 32 columns + 4 of padding
#define MAS_SLOT_F(R) (8 * (R) * MAS_CHUNK_F)

// v_writelane_b32: lane `lane` (uniform) of `old` := the scalar `val`
static __device__ __forceinline__ int mas_writelane(int val, int lane, int old)
{
    // (two scalar registers in one vector instruction exceed gfx9's constant bus: the lane select goes through m0)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                 : "+v"(old)
                 : "s"(__builtin_amdgcn_readfirstlane(val)), "s"(__builtin_amdgcn_readfirstlane(lane))
                 : "m0");
    return old;
}

// the loader's view of the two progress counters: ds_ instructions the compiler does not see (for its own ds_read / ds_write it drains
// the wave's LDS-DMAs first -- it cannot tell that they go elsewhere)
static __device__ __forceinline__ int mas_lds_peek(const int* p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(uintptr_t)(const __attribute__((address_space(3))) int*)p) : "memory");
    return v;
}
static __device__ __forceinline__ void mas_lds_poke(int* p, int v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(uintptr_t)(__attribute__((address_space(3))) int*)p), "v"(v) : "memory");
}
template <int N> static __device__ __forceinline__ void mas_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

#define MAS_OUT_RING 8        // blocks of a wave's last row kept for the wave below (it runs at most MAS_RING blocks ahead of it)

// one 32-column block of a DP wave (one row per lane).  MODE 0: all 32 columns exist and none is column 0 (no per-column tests);
// 1: the first block of a lattice of >= 32 columns (column 0 is the only special one -- every wave's start waits for the first block of
// the wave above, so this block is the pipeline's fill); 2: anything (tests per column).
// in[j] = the row above the wave's first row at column j - 1 of the block (in[0]: the previous block's last column).
// U = the row above the wave's first row: inp[q] = U[4q .. 4q+3] (the columns of this block), up_m1 = U[-1] (the previous block's last
// column).  Column j needs U[j - 1]: the previous group's .w for the first column of a group -- a register rename, no shifted copy of U
// is ever made.  cur / icur = the first groups of the lane's row and of U, already read (the block's combined poll fetched them).
// lastp[j] <- the lane's new value of column j (lane 63's pointer is the hand-over row for the wave below, the others' a scratch row).
template <bool TIE_MOVE, int MODE>
static __device__ __forceinline__ float mas_dp_block(const float4* __restrict__ rowp, const float4* __restrict__ inp, float4 cur, float4 icur,
                                                     float up_m1, int blk, int y_len, int ngroup, bool first_row_here, float& prev,
                                                     unsigned& bits, float* __restrict__ lastp)
{
    float iw_prev = up_m1;
#if __HIP_DEVICE_COMPILE__
    float4 nxt = cur, inxt = icur;                         // (the next group's values are read a group early: LDS latency)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int g = blk * 8 + q;
        if (MODE == 2 && g >= ngroup) break;
        if (q < 7) {
            nxt = rowp[q + 1];
            inxt = inp[q + 1];
        }
#pragma unroll
        for (int j4 = 0; j4 < 4; ++j4) {
            const int y = 4 * g + j4;
            if (MODE == 2 && y >= y_len) break;
            const int j = q * 4 + j4;
            if ((MODE == 1 && j == 0) || (MODE == 2 && y == 0)) {
                prev = first_row_here ? cur.x : MAS_NEG;
            } else {
                // the row above is lane - 1's (DPP wave_shr:1); lane 0 keeps `old`: the wave / band above, read from LDS by every lane
                const float a = prev;
                const float s_up = (j4 == 0) ? iw_prev : (j4 == 1) ? icur.x : (j4 == 2) ? icur.y : icur.z;
                const float cc = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(s_up), __float_as_int(prev), 0x138, 0xf, 0xf, false));
                // v1: direction = where(a > c, 0, -1);  v2/Triton: move iff c > a.  max = where(a > c, a, c).
                const bool move = TIE_MOVE ? !(a > cc) : (cc > a);
                // bits = 2 bits + move in one instruction (add with the compare's lane mask as carry-in): column j of the block ends at
                // bit 31 - j.  max(a, c) as v_max_f32 rather than a select on the compare: the same value (for equal operands either
                // one; the sign of a zero cannot change a later compare), one dependent instruction less.  One fp32 add per cell.
                unsigned long long carry_out;
                float m;
                asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(bits), "=s"(carry_out) : "v"(bits), "s"(__builtin_amdgcn_ballot_w64(move)));
                asm("v_max_f32_e32 %0, %1, %2" : "=v"(m) : "v"(a), "v"(cc));
                const float val = (j4 == 0) ? cur.x : (j4 == 1) ? cur.y : (j4 == 2) ? cur.z : cur.w;
                asm("v_add_f32_e32 %0, %1, %2" : "=v"(prev) : "v"(val), "v"(m));
            }
            lastp[j] = prev;                               // (every lane writes: one ds_write, no lane select; lane 63's lands in the hand-over row)
        }
        iw_prev = icur.w;
        cur = nxt;
        icur = inxt;
    }
#endif
    return iw_prev;                                        // U[31]: the next block's U[-1]
}

template <int R>
static __device__ __forceinline__ void mas_backtrack_wave(int b, int lane, int x_len, int y_len, int Tx, int Ty, int P, int nblk_max,
                                                          const unsigned* __restrict__ masks, float* __restrict__ path, int* __restrict__ dur,
                                                          int* __restrict__ rows);

#define MAS_NL 2              // loader waves per band: ONE wave's LDS-DMA stream lands a 16 KB block every ~0.65 us (MI355X_MICROARCH.md,
                              // ldsdma-fill) -- exactly what a DP wave needs for its 32 columns, so one loader paced the whole lattice
template <int W, bool TIE_MOVE>
__global__ void __launch_bounds__(64 * (W + MAS_NL))
mas_band_kernel(const float* __restrict__ value, const int* __restrict__ t_x, const int* __restrict__ t_y, int B, int Tx, int Ty, int P,
                int nblk_max, u64* __restrict__ xchg, unsigned* __restrict__ masks, int n_band_wgs, float* __restrict__ path,
                int* __restrict__ dur, int* __restrict__ rows, unsigned* __restrict__ status, int fused)
{
#if __HIP_DEVICE_COMPILE__
    if ((int)blockIdx.x >= n_band_wgs) {
        // the workgroups behind the bands clear the outputs on the CUs the DP leaves idle (the backtrack launch that follows writes the
        // path's ones, the row indices and the durations): no memset launches in front of the DP
        const size_t z = blockIdx.x - n_band_wgs, nz = gridDim.x - n_band_wgs, nt = (size_t)64 * (W + MAS_NL);
        const size_t n_path = path ? (size_t)B * Tx * Ty : 0, n_dur = dur ? (size_t)B * Tx : 0, n_rows = rows ? (size_t)B * Ty : 0;
        for (size_t i = z * nt + threadIdx.x; i < n_path; i += nz * nt) path[i] = 0.f;
        for (size_t i = z * nt + threadIdx.x; i < n_dur; i += nz * nt) dur[i] = 0;
        for (size_t i = z * nt + threadIdx.x; i < n_rows; i += nz * nt) rows[i] = -1;
        return;
    }
    // a ring slot = one 32-column block of the band's 64 W rows as 8 W chunks of 8 rows x 128 bytes (one DMA instruction each: lane =
    // (row of the chunk, 16-byte piece), i.e. whole lines -- with a lane per ROW an instruction touches 64 lines, and the texture path
    // needs ~4 cycles for each: that, not the DP, was the 0.4 us per column of the one-workgroup kernel).  16 bytes of padding per
    // chunk spread the DP's reads (lane = row, stride 128 bytes) over the banks.
    __shared__ __attribute__((aligned(16))) float ring[MAS_RING * MAS_SLOT_F(W)];
    __shared__ int ctr[MAS_NL + W];                        // [l] blocks loaded by loader l, [MAS_NL + w] blocks finished (consumed AND last row published) by wave w
    __shared__ __attribute__((aligned(16))) float outrow[W][MAS_OUT_RING][MAS_BLK];   // lane 63's values of the last blocks: the row above the wave below
    __shared__ float scratch[W][64 * 33];                  // where the other lanes' per-column writes go (33: conflict-free)
    __shared__ __attribute__((aligned(16))) float uprow[2][MAS_BLK];   // band head: the band above's last row, this block's and the next's
    __shared__ __attribute__((aligned(16))) float negrow[MAS_BLK];     // top band head: no row above
    // workgroups n, n+8, n+16, ... share an XCD: utterance b lives on XCD b % 8, its bands in dispatch order
    const int n = blockIdx.x, k = n >> 3;
    const int b = (k / P) * 8 + (n & 7), p = k % P;
    if (b >= B) return;
    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    const int band0 = p * 64 * W;
    if (x_len <= 0 || y_len <= 0 || band0 >= x_len) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < MAS_NL + W) ctr[tid] = 0;
    if (tid < MAS_BLK) negrow[tid] = MAS_NEG;
    __syncthreads();
    const int nblk = (y_len + MAS_BLK - 1) / MAS_BLK;
    const int ngroup = (y_len + 3) >> 2;

    if (wave >= W) {
        // ---- loader l of MAS_NL: instruction i of a block fetches rows band0 + 8 i .. + 7 (i = l, l + MAS_NL, ...), lane = (row, 16-byte piece) ----
        const int l = wave - W;
        constexpr int NI = 8 * W / MAS_NL;                                 // DMA instructions per block and loader
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(value + (size_t)b * Tx * Ty), 0,
                                                                            (int)((unsigned)Tx * (unsigned)Ty * 4u), 0x00020000);
        unsigned rowoff[NI];
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int x = band0 + 8 * (l + MAS_NL * k) + (lane >> 3);
            rowoff[k] = x < x_len ? (unsigned)x * (unsigned)Ty * 4u : 0xFFFFFF00u;       // outside the descriptor: reads zero
        }
        for (int blk = 0; blk < nblk + MAS_AHEAD; ++blk) {
            if (blk < nblk) {
                while (true) {                             // the slot's previous block consumed by every DP wave
                    int c = mas_lds_peek(&ctr[MAS_NL]);
#pragma unroll
                    for (int w = 1; w < W; ++w) {
                        const int cw = mas_lds_peek(&ctr[MAS_NL + w]);
                        c = cw < c ? cw : c;
                    }
                    if (c >= blk - (MAS_RING - 1)) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                const int slot = blk % MAS_RING;
                const unsigned col = (unsigned)(blk * 8 + (lane & 7)) * 16u;
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    const unsigned voff = (col < (unsigned)Ty * 4u && rowoff[k] != 0xFFFFFF00u) ? rowoff[k] + col : 0xFFFFFF00u;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (mas_lds_void*)(ring + slot * MAS_SLOT_F(W) + (l + MAS_NL * k) * MAS_CHUNK_F), 16, voff, 0, 0, 0);
                }
            }
            if (blk >= MAS_AHEAD) {
                if (blk < nblk) mas_wait_vmcnt<MAS_AHEAD * NI>();
                else mas_wait_vmcnt<0>();
                if (lane == 0) mas_lds_poke(&ctr[l], blk - MAS_AHEAD + 1);
            }
        }
        return;
    }

    // ---- DP wave `wave`: rows band0 + 64 wave + lane ----
    const int w = wave, x_local = 64 * w + lane;
    if (band0 + 64 * w >= x_len) {                         // no rows: never holds the ring back
        if (lane == 0) __hip_atomic_store(&ctr[MAS_NL + w], 0x7fffffff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }
    const bool from_wave = w > 0, from_band = w == 0 && p > 0;
    const bool to_band = w + 1 == W && p + 1 < P && band0 + 64 * W < x_len;
    const int Typ = nblk_max * MAS_BLK;
    u64* xo = xchg + (size_t)(b * P + p) * Typ;
    const u64* xi = xchg + (size_t)(b * P + p - 1) * Typ;
    unsigned* mo = masks + (size_t)(b * P + p) * nblk_max * W * 64;
    float prev = MAS_NEG, up_m1 = MAS_NEG;
    auto xcol = [&](int blk) {
        const int col = blk * MAS_BLK + (lane & 31);
        return col < y_len ? col : y_len - 1;
    };
    // band head: the last row of the band above arrives as 8-byte {value, tag} words (written once per launch); a block's words are
    // requested a block early (vnext) and put into uprow a block early too, so that neither the round trip to L2 nor the LDS write ->
    // read sits between two blocks of the DP
    u64 vnext = 0;
    auto fetch_up = [&](int blk) {
        u64 word = vnext;
        const u64* wp = xi + xcol(blk);
        for (int spin = 0; (unsigned)(word >> 32) != 1u && spin < (1 << 22); ++spin) {
            __builtin_amdgcn_s_sleep(1);
            word = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // gave up (the band above never got a CU, or died): what follows is not the alignment -- say so (as_device_status)
        if ((unsigned)(word >> 32) != 1u) as_status_raise(status, AS_STATUS_MAS_TIMEOUT);
        if (blk + 1 < nblk) vnext = __hip_atomic_load(xi + xcol(blk + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < MAS_BLK) uprow[blk & 1][lane] = __uint_as_float((unsigned)word);
    };
    if (from_band) {
        vnext = __hip_atomic_load(xi + xcol(0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fetch_up(0);
    }
    auto lds_addr = [](const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p; };
    static_assert(MAS_NL == 2, "the block's poll reads the two loaders' counters");
    const unsigned a_c0 = lds_addr(&ctr[0]), a_c1 = lds_addr(&ctr[1]), a_cw = from_wave ? lds_addr(&ctr[MAS_NL + w - 1]) : a_c0;
#ifdef AS_EXPERIMENTS
    long long c_wait = 0, c_dp = 0, c_all = clock64();
#define MAS_T(v) const long long v = clock64()
#define MAS_ACC(acc, a, b_) acc += (b_) - (a)
#else
#define MAS_T(v)
#define MAS_ACC(acc, a, b_)
#endif
    for (int blk = 0; blk < nblk; ++blk) {
        MAS_T(t0);
        const float* slot = ring + (blk % MAS_RING) * MAS_SLOT_F(W);
        const float4* rowp = reinterpret_cast<const float4*>(slot + (x_local >> 3) * MAS_CHUNK_F + (x_local & 7) * 32);
        const float4* inp = reinterpret_cast<const float4*>(from_wave ? &outrow[w - 1][blk % MAS_OUT_RING][0] : (from_band ? &uprow[blk & 1][0] : &negrow[0]));
        // ONE LDS round trip per block: the loader's counter, the counter of the wave above and -- speculatively, behind them in the
        // queue -- the block's first operands.  The LDS serves a wave's requests in order, and both producers write their data before
        // their counter: if the counters say "there", the operands read after them are the block's.
        float4 cur0, i0;
        while (true) {
            int c0, c1, c2;
            asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %7\n\tds_read_b128 %3, %8\n\tds_read_b128 %4, %9\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(cur0), "=&v"(i0)
                         : "v"(a_c0), "v"(a_c1), "v"(a_cw), "v"(lds_addr(rowp)), "v"(lds_addr(inp))
                         : "memory");
            if (__builtin_amdgcn_readfirstlane(c0) > blk && __builtin_amdgcn_readfirstlane(c1) > blk && __builtin_amdgcn_readfirstlane(c2) > blk) break;
            __builtin_amdgcn_s_sleep(1);
        }
        MAS_T(t1);
        MAS_ACC(c_wait, t0, t1);
#ifdef AS_EXPERIMENTS
        if (b == 0 && lane == 0 && blk == 0) mas_dbg2[(p * W + w) * 4 + 0] = wall_clock64();
#endif
        float* lastp = lane == 63 ? &outrow[w][blk % MAS_OUT_RING][0] : &scratch[w][lane * 33];
        unsigned bits = 0u;
        const int ncol = (y_len - blk * MAS_BLK) < MAS_BLK ? (y_len - blk * MAS_BLK) : MAS_BLK;
        if (blk > 0 && ncol == MAS_BLK) {
            up_m1 = mas_dp_block<TIE_MOVE, 0>(rowp, inp, cur0, i0, up_m1, blk, y_len, ngroup, false, prev, bits, lastp);
        } else if (blk == 0 && ncol == MAS_BLK) {
            up_m1 = mas_dp_block<TIE_MOVE, 1>(rowp, inp, cur0, i0, up_m1, blk, y_len, ngroup, band0 + x_local == 0, prev, bits, lastp);
        } else {
            up_m1 = mas_dp_block<TIE_MOVE, 2>(rowp, inp, cur0, i0, up_m1, blk, y_len, ngroup, band0 + x_local == 0, prev, bits, lastp);
            const int have = blk == 0 ? ncol - 1 : ncol;   // decisions shifted in (column 0 has none)
            bits = have > 0 ? bits << (MAS_BLK - ncol) : 0u;
        }
        MAS_T(t3);
        MAS_ACC(c_dp, t1, t3);
#ifdef AS_EXPERIMENTS
        if (b == 0 && lane == 0 && blk == 0) mas_dbg2[(p * W + w) * 4 + 1] = wall_clock64();
        if (b == 0 && lane == 0 && blk == nblk - 1) mas_dbg2[(p * W + w) * 4 + 2] = wall_clock64();
#endif
        mo[((size_t)blk * W + w) * 64 + lane] = bits;
        if (to_band && lane < ncol)
            __hip_atomic_store(xo + blk * MAS_BLK + lane, ((u64)1u << 32) | __float_as_uint(outrow[w][blk % MAS_OUT_RING][lane]), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // finished: the loader may reuse the slot, the wave below may read this block's last row (release: the row's ds_writes first)
        if (lane == 0) __hip_atomic_store(&ctr[MAS_NL + w], blk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (from_band && blk + 1 < nblk) fetch_up(blk + 1);
    }
    if (fused) {
        // Small lattices (one band, a few thousand cells): the whole search is ONE launch.  The band's own DP waves clear the utterance's
        // outputs (no clearing workgroups, no second launch), wait for each other -- every wave's decision words and zeros are in memory
        // (vmcnt(0)) before the barrier -- and the first wave walks the path back, block after block.  [32,40,100]: two launches were 26 us.
        const int nlive = min(W, (x_len - band0 + 63) >> 6);                // DP waves that have rows (the others and the loader have exited)
        const int t = 64 * w + lane, nt = 64 * nlive;
        if (path) for (int i = t; i < Tx * Ty; i += nt) path[(size_t)b * Tx * Ty + i] = 0.f;
        if (dur) for (int i = t; i < Tx; i += nt) dur[(size_t)b * Tx + i] = 0;
        if (rows) for (int i = t; i < Ty; i += nt) rows[(size_t)b * Ty + i] = -1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (w == 0) mas_backtrack_wave<W>(b, lane, x_len, y_len, Tx, Ty, P, nblk_max, masks, path, dur, rows);
    }
#ifdef AS_EXPERIMENTS
    if (b == 0 && p == 0 && tid == 0) {                      // the first wave of the first band of utterance 0
        mas_dbg[0] = c_wait; mas_dbg[1] = 0; mas_dbg[2] = c_dp; mas_dbg[3] = clock64() - c_all;
    }
#endif
#endif
}

// The walk through ONE block of 32 columns.  Lane lambda of the wave holds the word W = the 32 decisions of row base - lambda; the path enters
// the block (at its top column j_top) in row idx = base - off and leaves it d rows higher (returned; T = which steps moved).  A step is
// a shift and an add in scalar registers (a full block: 32 lane masks made up front, then s_bfe / s_lshl1_add / s_add per column on
// static registers).
static __device__ __forceinline__ int mas_walk_steps(unsigned W, int off, int j_top, int j_low, unsigned& T)
{
    int d = 0;
    T = 0u;                                                // bit k: the step at column j_top - k moved up
    if (j_top == 31 && j_low == 0) {
        // a full block.  M[j] = the lane mask of "row (base - lane) moves up at column j" (a compare per column, all 32 up front):
        // the walk is then three scalar instructions per column on static registers -- shift M[j] by the current lane, and, add.
        u64 M[MAS_BLK];
#pragma unroll
        for (int j = 0; j < MAS_BLK; ++j) M[j] = __builtin_amdgcn_ballot_w64(((W >> (31 - j)) & 1u) != 0u);
        // three scalar instructions per column: s_bfe_u64 takes the bit of M[j] at the current lane (offset in the low bits of its second
        // operand, width 1 in bits 16..22 -- the running lane number carries the width field along), s_lshl1_add_u32 shifts it into T,
        // s_add moves the lane on.  T collects the steps MSB first; reversed afterwards (bit k = the step at column 31 - k).
        unsigned lamw = (unsigned)__builtin_amdgcn_readfirstlane(off) | (1u << 16);
#pragma unroll
        for (int j = 31; j >= 0; --j) {
            u64 bb;
            asm("s_bfe_u64 %0, %1, %2" : "=s"(bb) : "s"(M[j]), "s"(lamw));
            const unsigned bit = (unsigned)bb;
            asm("s_lshl1_add_u32 %0, %1, %2" : "=s"(T) : "s"(T), "s"(bit));
            lamw += bit;
        }
        T = __builtin_bitreverse32(T);
        const int lam = (int)(lamw & 0xFFFFu);
        d = lam - off;
    } else {
        for (int j = j_top; j >= j_low; --j) {
            const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)W, __builtin_amdgcn_readfirstlane(off + d));
            const unsigned bit = (w >> (31 - j)) & 1u;
            T |= bit << (j_top - j);
            d += (int)bit;
        }
    }
    return __builtin_amdgcn_readfirstlane(d);
}
// lane = column of the block: the row of column y - 1, the path cell, the duration count
static __device__ __forceinline__ void mas_walk_emit(unsigned T, int idx, int blk, int j_top, int j_low, int b, int Tx, int Ty,
                                                     float* __restrict__ path, int* __restrict__ dur, int* __restrict__ rows, int lane)
{
    if (lane >= j_low && lane <= j_top) {
        const int kk = j_top - lane;
        const int rr = idx - __popc(T & (0xFFFFFFFFu >> (31 - kk)));
        const int col = blk * MAS_BLK + lane - 1;
        if (path) path[((size_t)b * Tx + rr) * Ty + col] = 1.f;
        if (rows) rows[(size_t)b * Ty + col] = rr;
        if (dur) atomicAdd(dur + (size_t)b * Tx + rr, 1);
    }
}

// word of row `row` in block `blk` of utterance b (R = DP waves per band: wave i / 64, lane i % 64 of band row / (64 R))
template <int R>
static __device__ __forceinline__ const unsigned* mas_word_ptr(const unsigned* __restrict__ masks, int b, int P, int nblk_max, int row, int blk)
{
    const int pp = row / (64 * R), i = row - pp * 64 * R;
    return masks + (((size_t)(b * P + pp) * nblk_max + blk) * R + (i >> 6)) * 64 + (i & 63);
}

// The whole walk by ONE wave, block after block from the last column down (small lattices, inside the DP launch: mas_band_kernel's fused
// tail).  The words of the next block are fetched while this one is walked (the row can only have moved up by 32 by then: 64 lanes cover it).
template <int R>
static __device__ __forceinline__ void mas_backtrack_wave(int b, int lane, int x_len, int y_len, int Tx, int Ty, int P, int nblk_max,
                                                          const unsigned* __restrict__ masks, float* __restrict__ path, int* __restrict__ dur,
                                                          int* __restrict__ rows)
{
    int idx = x_len - 1;
    if (lane == 0) {
        if (path) path[((size_t)b * Tx + idx) * Ty + (y_len - 1)] = 1.f;
        if (rows) rows[(size_t)b * Ty + (y_len - 1)] = idx;
        if (dur) atomicAdd(dur + (size_t)b * Tx + idx, 1);
    }
    if (y_len < 2) return;
    // decisions of rows base - lane in block blk (row 0 never moves, rows above the lattice do not exist: zero).  The load is issued
    // here and awaited by arrive: the compiler would wait for it at the first v_readlane of the walk it is meant to overlap.
    auto issue = [&](unsigned& dst, int base, int blk) -> bool {
        const int row = base - lane;
        const bool ok = row > 0 && blk >= 0;
        const unsigned* ptr = mas_word_ptr<R>(masks, b, P, nblk_max, ok ? row : 0, ok ? blk : 0);
        asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
        return ok;
    };
    auto arrive = [&](unsigned& dst, bool ok) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(dst) : : "memory");
        dst = ok ? dst : 0u;
    };
    int blk = (y_len - 1) >> 5;
    int base = idx;
    unsigned W, Wn;
    const bool ok0 = issue(W, base, blk);
    arrive(W, ok0);
    while (blk >= 0) {
        const bool okn = issue(Wn, idx, blk - 1);          // in flight during the walk
        const int j_top = (blk == (y_len - 1) >> 5) ? ((y_len - 1) & 31) : 31;
        const int j_low = blk == 0 ? 1 : 0;
        unsigned T;
        const int d = mas_walk_steps(W, base - idx, j_top, j_low, T);
        arrive(Wn, okn);                                   // (before this block's stores are issued: vmcnt(0) would wait for them too)
        mas_walk_emit(T, idx, blk, j_top, j_low, b, Tx, Ty, path, dur, rows, lane);
        base = idx;
        idx -= d;
        W = Wn;
        --blk;
    }
}

// ---- the parallel backtrack (lattices of more than one band, or too large to clear inside the DP's own workgroup) -----------------
// The walk is a chain of 2000 dependent steps only if it is run as one: a block's decisions define a FUNCTION "row at the block's top
// column -> row in front of its first column", and the path is the composition of the blocks' functions.
//   exit   (all blocks x all rows in parallel): row r's exit from block blk.  The path stays in a row until the highest column below
//          the current one whose decision bit is set in THAT row's word, then moves up one row: per move one LDS read at a
//          predictable address (the words of rows r, r-1, r-2, ...), a mask and a find-first-bit -- no dependent memory round trips.
//          The table holds r - exit(r) <= 32 in a byte.
//   chain  (one workgroup per utterance): the table in LDS (63 blocks x 1024 rows = 63 KB), the entry row of every block by 62
//          dependent LDS reads.
//   emit   (all blocks in parallel, one wave each): the existing per-block walk from the block's now known entry row; rows, path cells
//          and durations are written by the lanes.
// [8,1024,2000]: 40 us as one chain of blocks -> three launches of a few microseconds.
template <int R>
__global__ void __launch_bounds__(1024)
mas_exit_kernel(const int* __restrict__ t_x, const int* __restrict__ t_y, int Tx, int Ty, int P, int nblk_max, const unsigned* __restrict__ masks,
                unsigned char* __restrict__ ex, int exs)
{
    extern __shared__ unsigned words[];                    // [4 + x_len] decision words of this block (four zero words in front: rows < 0)
    const int blk = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    if (x_len <= 0 || y_len < 2 || blk > ((y_len - 1) >> 5)) return;
    if (tid < 4) words[tid] = 0u;
    for (int r = tid; r < x_len; r += 1024) words[4 + r] = r > 0 ? *mas_word_ptr<R>(masks, b, P, nblk_max, r, blk) : 0u;      // (row 0 never moves)
    __syncthreads();
    const int j_top = (blk == (y_len - 1) >> 5) ? ((y_len - 1) & 31) : 31;
    const int j_low = blk == 0 ? 1 : 0;
    // decision of column j = bit 31 - j: the columns j_low .. j_top are the bits 31 - j_top .. 31 - j_low
    const unsigned valid = (0xFFFFFFFFu >> j_low) & (0xFFFFFFFFu << (31 - j_top));
    unsigned char* out = ex + ((size_t)b * nblk_max + blk) * exs;      // (row stride exs: Tx rounded up to 16 bytes)
    for (int r = tid; r < x_len; r += 1024) {
        // the words of rows cur, cur - 1, cur - 2, cur - 3 are held ahead of their use: the addresses do not depend on the data, only
        // the number of moves does, so the chain per move is a mask, a find-first-bit and a shift -- the LDS latency is three moves away
        const unsigned* wp = words + 4 + r;
        unsigned w0 = wp[0], w1 = wp[-1], w2 = wp[-2], w3 = wp[-3];
        int moves = 0;
        unsigned m = valid;                                // columns not yet passed
        while (true) {
            const unsigned w = w0 & m;
            if (w == 0u) break;                            // stays in this row down to the block's first column
            const int pbit = __builtin_ctz(w);             // the HIGHEST column with a set bit = the lowest bit index
            ++moves;
            if (pbit >= 31) break;
            m = (0xFFFFFFFFu << (pbit + 1)) & valid;       // on with the columns below it (higher bit indices)
            w0 = w1; w1 = w2; w2 = w3;
            w3 = (r - moves - 3 >= -4) ? wp[-moves - 3] : 0u;
        }
        out[r] = (unsigned char)moves;
    }
}

__global__ void __launch_bounds__(1024)
mas_chain_kernel(const int* __restrict__ t_x, const int* __restrict__ t_y, int Tx, int Ty, int nblk_max, const unsigned char* __restrict__ ex,
                 int exs, int* __restrict__ entry, int lds_bytes)
{
    extern __shared__ unsigned char tab[];                 // a chunk of blocks: [blocks][rowsP]
    const int b = blockIdx.x, tid = threadIdx.x;
    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    if (x_len <= 0 || y_len <= 0) return;
    const int last = (y_len - 1) >> 5;
    const int rowsP = (x_len + 15) & ~15;                  // 16-byte rows of the LDS copy
    const int per = lds_bytes / rowsP;                     // blocks per chunk (>= 1: the host checks)
    __shared__ int cur_row;
    if (tid == 0) { cur_row = x_len - 1; entry[(size_t)b * nblk_max + last] = x_len - 1; }
    for (int hi = last; hi >= 1; hi -= per) {              // blocks hi, hi - 1, ..., lo of this chunk (block 0's exit is not needed)
        const int lo = hi - per + 1 > 1 ? hi - per + 1 : 1;
        const int nb = hi - lo + 1;
        __syncthreads();
        // 16 bytes per load, whole rows of the table (its row stride is a multiple of 16; rows past x_len are never looked up)
        const int vec_per_row = rowsP >> 4;
        for (int i = tid; i < nb * vec_per_row; i += 1024) {
            const int k = i / vec_per_row, v = i - k * vec_per_row;
            reinterpret_cast<uint4*>(tab)[(size_t)k * vec_per_row + v] =
                reinterpret_cast<const uint4*>(ex + ((size_t)b * nblk_max + lo + k) * exs)[v];
        }
        __syncthreads();
        if (tid == 0) {
            int r = cur_row;
            for (int blk = hi; blk >= lo; --blk) {
                r -= tab[(size_t)(blk - lo) * rowsP + r];
                entry[(size_t)b * nblk_max + blk - 1] = r;
            }
            cur_row = r;
        }
    }
}

template <int R>
__global__ void __launch_bounds__(64)
mas_emit_kernel(const int* __restrict__ t_x, const int* __restrict__ t_y, int Tx, int Ty, int P, int nblk_max, const unsigned* __restrict__ masks,
                const int* __restrict__ entry, float* __restrict__ path, int* __restrict__ dur, int* __restrict__ rows)
{
    const int blk = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    int x_len = t_x[b], y_len = t_y[b];
    x_len = x_len > Tx ? Tx : x_len;
    y_len = y_len > Ty ? Ty : y_len;
    if (x_len <= 0 || y_len <= 0) return;
    const int last = (y_len - 1) >> 5;
    if (blk > last) return;
    if (blk == last && lane == 0) {                        // the path's last cell
        const int idx = x_len - 1;
        if (path) path[((size_t)b * Tx + idx) * Ty + (y_len - 1)] = 1.f;
        if (rows) rows[(size_t)b * Ty + (y_len - 1)] = idx;
        if (dur) atomicAdd(dur + (size_t)b * Tx + idx, 1);
    }
    if (y_len < 2) return;
    const int idx = entry[(size_t)b * nblk_max + blk];
    const int row = idx - lane;
    const unsigned W = row > 0 ? *mas_word_ptr<R>(masks, b, P, nblk_max, row, blk) : 0u;
    const int j_top = blk == last ? ((y_len - 1) & 31) : 31;
    const int j_low = blk == 0 ? 1 : 0;
    if (j_top < j_low) return;                             // (a one-column last block that is also block 0)
    unsigned T;
    (void)mas_walk_steps(W, 0, j_top, j_low, T);
    mas_walk_emit(T, idx, blk, j_top, j_low, b, Tx, Ty, path, dur, rows, lane);
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
    if (w > (r <= 1 ? 16 : 8)) return AS_EINVAL;           // Tx <= 8192 (the register budget of mas_launch)
    *R = r;
    *W = w;
    return AS_OK;
}

// banded variant: R DP waves of 64 rows per band (1 up to 64 rows, else 2; four waves per band were measured: each wave runs a block
// behind the one above, so the ring must hold W + 3 blocks or the first wave starves -- 0.28 ms against 0.19), P bands, workspace = exchange words + decision words
struct MasBands {
    int R, P, nblk, exs;
    bool fused;                // one launch: DP, clearing and the walk inside the band's workgroup
    size_t xchg_bytes, mask_bytes, ex_bytes, entry_bytes;
};
#define MAS_FUSED_CELLS 16384  /* Tx * Ty up to which the band's own waves clear the dense path (64 stores per lane at most) */
#define MAS_CHAIN_LDS (128 * 1024)
static MasBands mas_bands(int B, int Tx, int Ty)
{
    MasBands g;
    g.R = Tx <= 64 ? 1 : 2;
    g.P = as_cdiv(Tx, 64 * g.R);
    g.nblk = as_cdiv(Ty, MAS_BLK);
    g.xchg_bytes = g.P > 1 ? (((size_t)B * g.P * g.nblk * MAS_BLK * sizeof(u64) + 255) & ~(size_t)255) : 0;
    g.mask_bytes = (((size_t)B * g.P * g.nblk * g.R * 64 * sizeof(unsigned)) + 255) & ~(size_t)255;
    g.fused = g.P == 1 && (long)Tx * Ty <= MAS_FUSED_CELLS && !getenv("AS_MAS_NO_FUSE");
    g.exs = (Tx + 15) & ~15;
    g.ex_bytes = g.fused ? 0 : (((size_t)B * g.nblk * g.exs) + 255) & ~(size_t)255;       // exit table: a byte per (block, row)
    g.entry_bytes = g.fused ? 0 : (((size_t)B * g.nblk * sizeof(int)) + 255) & ~(size_t)255;
    return g;
}
static bool mas_banded_ok(const float* value, int Tx, int Ty)
{
    const char* env = getenv("AS_MAS_IMPL");               // tuning/experiments only: "one" = the one-workgroup kernel
    return !(env && !strcmp(env, "one")) && Ty % 4 == 0 && (reinterpret_cast<uintptr_t>(value) & 15) == 0 && (double)Tx * Ty * 4.0 < 2147483648.0;
}

extern "C" size_t as_mas_workspace_bytes(int B, int Tx, int Ty)
{
    int R, W;
    if (B <= 0 || Tx <= 0 || Ty <= 0) return 0;
    const MasBands g = mas_bands(B, Tx, Ty);
    const size_t banded = g.xchg_bytes + g.mask_bytes + g.ex_bytes + g.entry_bytes;
    // rows that are not 16-byte aligned take the one-workgroup kernel (Tx <= 8192): the caller's buffer serves either
    const size_t one = mas_geometry(Tx, &R, &W) == AS_OK ? (size_t)B * ((Ty + 3) / 4) * 64 * W * sizeof(u64) : 0;
    if (one == 0) return 0;
    return banded > one ? banded : one;
}

template <int R, bool TIE>
static void mas_launch(bool vec4, int B, int W, size_t smem, hipStream_t s, const float* value, const int* t_x,
                       const int* t_y, int Tx, int Ty, float* path, int* dur, int* rows, u64* ws, int sc)
{
    // 16-byte pieces of a row in flight per lane (see load_chunk)
    constexpr int Q = 1;      // (whole 128-byte lines per row, Q = 8, measured SLOWER here: 1.04 against 0.80 ms -- the same 64 lines per instruction)
    // W == 1 (Tx <= 64 R): a 64-thread workgroup may use the whole register file; 16 waves (R = 1) leave 128 VGPRs each, 8 waves 256.
    if (W == 1) {
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, Q, true, TIE, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
        else
            hipLaunchKernelGGL((mas_kernel<R, Q, false, TIE, 64>), dim3(B), dim3(64), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
    } else {
        constexpr int MT = R <= 1 ? 1024 : 512;
        if (vec4)
            hipLaunchKernelGGL((mas_kernel<R, Q, true, TIE, MT>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
        else
            hipLaunchKernelGGL((mas_kernel<R, Q, false, TIE, MT>), dim3(B), dim3(64 * W), smem, s, value, t_x, t_y, Tx, Ty, path, dur, rows, ws, sc);
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
    if (mas_banded_ok(value, Tx, Ty)) {
        const MasBands g = mas_bands(B, Tx, Ty);
        if (!ws || ws_bytes < g.xchg_bytes + g.mask_bytes + g.ex_bytes + g.entry_bytes) return AS_EINVAL;
        u64* xchg = static_cast<u64*>(ws);
        unsigned char* wsb = static_cast<unsigned char*>(ws);
        unsigned* masks = reinterpret_cast<unsigned*>(wsb + g.xchg_bytes);
        unsigned char* ex = wsb + g.xchg_bytes + g.mask_bytes;
        int* entry = reinterpret_cast<int*>(wsb + g.xchg_bytes + g.mask_bytes + g.ex_bytes);
        if (g.xchg_bytes) AS_CHECK(hipMemsetAsync(xchg, 0, g.xchg_bytes, stream));
        AsProfScope prof__(AS_CLS_MAS, 2.0 * B * Tx * (double)Ty, 4.0 * B * Tx * (double)Ty * (path ? 2 : 1), stream);
        const int n_band_wgs = 8 * as_cdiv(B, 8) * g.P;
        const dim3 grid(n_band_wgs + (g.fused ? 0 : 512));  // + the workgroups that clear path / dur / rows (the fused form clears by itself)
        unsigned* status = as_status_words_device();
        const int fused = g.fused ? 1 : 0;
        if (g.R == 1) {
            if (tie_mode) hipLaunchKernelGGL((mas_band_kernel<1, true>), grid, dim3(64 * (1 + MAS_NL)), 0, stream, value, t_x, t_y, B, Tx, Ty, g.P, g.nblk, xchg, masks, n_band_wgs, path, dur, rows, status, fused);
            else hipLaunchKernelGGL((mas_band_kernel<1, false>), grid, dim3(64 * (1 + MAS_NL)), 0, stream, value, t_x, t_y, B, Tx, Ty, g.P, g.nblk, xchg, masks, n_band_wgs, path, dur, rows, status, fused);
        } else {
            if (tie_mode) hipLaunchKernelGGL((mas_band_kernel<2, true>), grid, dim3(64 * (2 + MAS_NL)), 0, stream, value, t_x, t_y, B, Tx, Ty, g.P, g.nblk, xchg, masks, n_band_wgs, path, dur, rows, status, fused);
            else hipLaunchKernelGGL((mas_band_kernel<2, false>), grid, dim3(64 * (2 + MAS_NL)), 0, stream, value, t_x, t_y, B, Tx, Ty, g.P, g.nblk, xchg, masks, n_band_wgs, path, dur, rows, status, fused);
        }
        if (!g.fused) {
            // the walk as three short launches (see above): every block's row -> row function, their composition, the emission
            const dim3 gb(g.nblk, B);
            const size_t lds_words = ((size_t)Tx + 4) * sizeof(unsigned);
            const int chain_lds = std::max(MAS_CHAIN_LDS / g.exs * g.exs, g.exs) > MAS_CHAIN_LDS ? g.exs : MAS_CHAIN_LDS;
            AS_LDS_OPT_IN(&mas_chain_kernel, MAS_CHAIN_LDS);
            if (g.R == 1) hipLaunchKernelGGL((mas_exit_kernel<1>), gb, dim3(1024), lds_words, stream, t_x, t_y, Tx, Ty, g.P, g.nblk, masks, ex, g.exs);
            else hipLaunchKernelGGL((mas_exit_kernel<2>), gb, dim3(1024), lds_words, stream, t_x, t_y, Tx, Ty, g.P, g.nblk, masks, ex, g.exs);
            hipLaunchKernelGGL(mas_chain_kernel, dim3(B), dim3(1024), chain_lds, stream, t_x, t_y, Tx, Ty, g.nblk, ex, g.exs, entry, chain_lds);
            if (g.R == 1) hipLaunchKernelGGL((mas_emit_kernel<1>), gb, dim3(64), 0, stream, t_x, t_y, Tx, Ty, g.P, g.nblk, masks, entry, path, dur, rows);
            else hipLaunchKernelGGL((mas_emit_kernel<2>), gb, dim3(64), 0, stream, t_x, t_y, Tx, Ty, g.P, g.nblk, masks, entry, path, dur, rows);
        }
        AS_CHECK_LAUNCH();
        return AS_OK;
    }
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
