// wave.hip.h — what every voice kernel shares (the fused kernels, the tile interpreter and the kernels specialised at run time,
// jit.cpp): which voices a wave owns, and the per-sample output stage — frames through buffer stores, the mix-down through an
// LDS transpose tile.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#include "kernel_args.hip.h"
#include "modules.hip.h"

#ifndef SRK_FRAME_AUX
#define SRK_FRAME_AUX 2  // cache policy of the frame stores: nt (write-once stream); tools/ builds variants with -DSRK_FRAME_AUX=
#endif


namespace srack {

namespace dev {

SRK_DEV double make_f64(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
SRK_DEV uint32_t f64_lo(double d) { return (uint32_t)__double2loint(d); }
SRK_DEV uint32_t f64_hi(double d) { return (uint32_t)__double2hiint(d); }

SRK_DEV float readlane_f32(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }  // `lane` wave-uniform

// Do the 32 samples of a control track that start at `p` hold one bit pattern?  (wave-uniform; one load and one compare per tile)
SRK_DEV bool track_flat(const float* p, int lane)
{
    const uint32_t v = ((const uint32_t*)p)[lane & 31];
    return __builtin_amdgcn_ballot_w64(v != (uint32_t)__builtin_amdgcn_readfirstlane((int)v)) == 0;
}

struct WaveMap {   // which voices a wave owns
    uint32_t wave0;     // first voice of the wave
    uint32_t n_active;  // real voices in it (lanes >= n_active shadow voice wave0 + n_active - 1: same work, same stores)
    uint32_t voice;     // this lane's voice (meaningful when active)
    uint32_t vc;        // this lane's voice clamped to a real one (safe to load from)
    bool active;
};

// Which group of `lanes` voices this workgroup owns.  (Tried: an XCD-aware map — workgroups are dealt to the 8 XCDs
// round-robin, so give XCD k the k-th contiguous eighth of the voices and let neighbouring 256-B pieces of a frame row
// leave through the same L2.  No measurable difference on the headline workload: 12.4 ms per step either way.)
template <class Args>
SRK_DEV uint32_t wave_index(const Args& a)
{
    return blockIdx.x - a.block0;
}

template <class Args>
SRK_DEV WaveMap wave_map(const Args& a, int lane)
{
    WaveMap m;
    m.wave0 = wave_index(a) * a.lanes;
    m.n_active = min(a.lanes, a.V - m.wave0);
    m.active = (uint32_t)lane < m.n_active;
    m.voice = m.wave0 + (uint32_t)lane;
    m.vc = m.active ? m.voice : m.wave0 + m.n_active - 1;
    return m;
}

}  // namespace dev

constexpr int kMixRows = 32;
static_assert(kMixRows == dev::kTileRows, "the tile-wise module forms (modules.hip.h) assume the mix tile's length");
constexpr int kMixPitch = 68;  // floats per LDS row of the mix tile: 64 lanes + 4 of padding (see emit_flush)
constexpr int kMixTile = kMixRows * kMixPitch;

// ---- per-sample output of the fused kernels ---------------------------------------------------------------
// kOut: 0 = decide at run time (exact-mode kernels), 1 = frames only, 2 = mix only, 3 = frames + mix.
// Frames: SGPR row base advanced by V per sample + a constant per-lane offset; lanes past V (only in the
// last wave) shadow voice V-1, compute the identical sample and store it to the identical address, so the
// store needs no exec mask.  Mix: the sample goes into a 32-row LDS tile; every 32 samples (and at the end)
// the rows are summed over the 64 lanes (emit_flush) and one lane per row writes the wave's partial.
struct Emit {
    float* frame_row;   // wave-uniform: this wave's 256 B of the current TILE's first frame row
    // Frames leave through a buffer store: descriptor (SGPRs, rebuilt per tile) + per-lane byte offset (a constant VGPR) +
    // scalar row offset advanced by V * 4 per sample — no vector arithmetic per store (a global_store needs a 64-bit
    // v_lshl_add per sample to form its address).
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t soff;      // byte offset of the current row inside the tile
    float* mp;          // wave-uniform: mixpart row of this wave
    bool has_frames, has_mix, full_wave;
    int lane, lane_c;
    uint32_t n_active;  // lanes of this wave that are real voices
};

template <int kOut>
SRK_DEV void emit_put(Emit& e, float* mix_tile, float o, int i, uint32_t V)  // i = row of the current 32-sample tile
{
    const bool frames = kOut == 0 ? e.has_frames : (kOut & 1) != 0;
    const bool mix = kOut == 0 ? e.has_mix : (kOut & 2) != 0;
    if (frames) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), e.rsrc, e.lane_c * 4, (int)e.soff, SRK_FRAME_AUX);
        e.soff += V * 4u;
    }
    if (mix) mix_tile[i * kMixPitch + e.lane] = o;  // (ds_write_addtid_b32 — no address VGPR — measured: no gain)
}

SRK_DEV void emit_rebase(Emit& e)  // point the descriptor at frame_row; a tile spans at most 32 rows = 32 * V * 4 bytes
{
    e.rsrc = __builtin_amdgcn_make_buffer_rsrc(e.frame_row, 0, 0x7fffffff, 0x00020000);
    e.soff = 0u;
}

// kBarrier = false: the caller's workgroup has other waves that do not take part (render_fm_pair_split); the tile is this wave's
// own, and one wave's LDS writes and reads stay in program order.
template <int kOut, bool kBarrier = true>
SRK_DEV void emit_flush(Emit& e, float* mix_tile, uint32_t t0, int n, uint32_t V)  // the tile holds samples t0 .. t0+n-1
{
    const bool frames = kOut == 0 ? e.has_frames : (kOut & 1) != 0;
    if (frames) {
        e.frame_row += (size_t)n * V;
        emit_rebase(e);
    }
    const bool mix = kOut == 0 ? e.has_mix : (kOut & 2) != 0;
    if (!mix) return;
    if (!e.full_wave && (uint32_t)e.lane >= e.n_active)  // shadow lanes contribute nothing to the mix
        for (int r = 0; r < kMixRows; r++) mix_tile[r * kMixPitch + e.lane] = 0.0f;
    if (kBarrier) __syncthreads();
    // lane l sums half (l >> 5) of row (l & 31) with eight 16-byte reads; row pitch 68 floats = 272 B keeps them 16-B aligned and
    // spreads the 16 lanes of a ds_read_b128 group (rows r .. r+15) over all 64 banks (bank = 4 r + 4 q mod 64)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* p = (const f4*)(mix_tile + (e.lane & 31) * kMixPitch + (e.lane >> 5) * 32);
    f4 acc = p[0];
#pragma unroll
    for (int q = 1; q < 8; q++) acc += p[q];
    float sum = (acc.x + acc.y) + (acc.z + acc.w);
    sum += __shfl_xor(sum, 32);
    if (e.lane < n) e.mp[t0 + e.lane] = sum;
    if (kBarrier) __syncthreads();
}

// Several planes, ONE tile (kernels generated for patches with two or more planes fed by wires, unless a tile-wise exact saw borrows a
// plane's tile — jit.cpp): a tile per plane is
// 8.7 KB of LDS each, and two of them already hold a CU to eight one-wave workgroups — two waves per SIMD, which leaves dependent
// arithmetic (P4's two powers per sample) half the issue slots (rocprofv3: VALU busy 50 % at two and at "four" waves per SIMD, the
// latter running in two rounds).  Plane j owns rows [j, j + 1) * kRows of the one tile, kRows = 32 / (planes, rounded up to a power of
// two); after the last plane's sample that fills them — every kRows samples — ONE pass sums all 32 rows exactly as emit_flush does (same
// reads, same additions: a plane's mix is bit for bit what its own tile gave) and lane r + j kRows stores row r of plane j.
template <int kOut>
SRK_DEV void emit_put_rows(Emit& e, float* rows, float o, int r, uint32_t V)  // rows = this plane's rows of the tile, r = i mod kRows
{
    const bool frames = kOut == 0 ? e.has_frames : (kOut & 1) != 0;
    const bool mix = kOut == 0 ? e.has_mix : (kOut & 2) != 0;
    if (frames) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o), e.rsrc, e.lane_c * 4, (int)e.soff, SRK_FRAME_AUX);
        e.soff += V * 4u;
    }
    if (mix) rows[r * kMixPitch + e.lane] = o;
}
template <int kRows>
SRK_DEV void emit_rows_flush(const Emit& e, float* tile, float* mp_lane, uint32_t t_first, int n_valid)  // mp_lane: the mixpart row of the plane lane & 31 belongs to (or null)
{
    static_assert(kRows == 16 || kRows == 8 || kRows == 4, "rows per plane");
    if (!e.full_wave && (uint32_t)e.lane >= e.n_active)
        for (int r = 0; r < kMixRows; r++) tile[r * kMixPitch + e.lane] = 0.0f;
    __syncthreads();
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4* p = (const f4*)(tile + (e.lane & 31) * kMixPitch + (e.lane >> 5) * 32);
    f4 acc = p[0];
#pragma unroll
    for (int q = 1; q < 8; q++) acc += p[q];
    float sum = (acc.x + acc.y) + (acc.z + acc.w);
    sum += __shfl_xor(sum, 32);
    const int r = e.lane & (kRows - 1);
    if (e.lane < 32 && r < n_valid && mp_lane) mp_lane[t_first + (uint32_t)r] = sum;
    __syncthreads();
}
template <int kOut>
SRK_DEV void emit_rows_end(Emit& e, int n, uint32_t V)  // end of a 32-sample tile: the mix is already out; the frame descriptor moves on
{
    const bool frames = kOut == 0 ? e.has_frames : (kOut & 1) != 0;
    if (frames) {
        e.frame_row += (size_t)n * V;
        emit_rebase(e);
    }
}

// A plane that carries a control track unchanged (every voice plays the same sample): the frames are a broadcast store of a
// wave-uniform value; the wave's mix partial is (number of real voices) x sample, written once per tile (emit_track_flush).
SRK_DEV void emit_track_put(Emit& e, float v, uint32_t V, bool frames)
{
    if (frames) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), e.rsrc, e.lane_c * 4, (int)e.soff, SRK_FRAME_AUX);
        e.soff += V * 4u;
    }
}
SRK_DEV void emit_track_flush(Emit& e, const float* track_t0, uint32_t t0, int n, uint32_t V, bool frames, bool mix)
{
    if (frames) {
        e.frame_row += (size_t)n * V;
        emit_rebase(e);
    }
    if (mix && e.lane < n) e.mp[t0 + e.lane] = (float)e.n_active * track_t0[e.lane];
}

SRK_DEV Emit make_emit(const KernelArgs& a, int plane, int lane)
{
    using dev::WaveMap;
    using dev::wave_map;
    Emit e;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t wave0 = wm.wave0;
    e.n_active = wm.n_active;
    e.full_wave = e.n_active == 64u;
    e.lane = lane;
    e.lane_c = min(lane, (int)e.n_active - 1);
    e.frame_row = a.frames ? a.frames + (size_t)plane * a.plane_stride + wave0 : nullptr;
    e.mp = a.mixpart ? a.mixpart + ((size_t)plane * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride : nullptr;
    e.has_frames = e.frame_row != nullptr;
    e.has_mix = e.mp != nullptr;
    emit_rebase(e);
    return e;
}

}  // namespace srack
