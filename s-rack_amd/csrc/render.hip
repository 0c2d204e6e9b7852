// render.hip — launch code of the batch-render path (gfx950 only): device copies of the flattened programs, the
// chunk / control-pipeline schedule of one render, kernel dispatch, timings and state read-back.
//
// Execution model: one voice per lane, 64 voices per wave, one wave per workgroup (no barriers between waves: voices
// never interact).  A wave owns its voices for the whole render because every module is a recurrence in time (phase
// accumulator, IIR state, envelope state).  The kernels live in interp.hip.h (generic tile interpreter) and
// fused.hip.h (whole patch shapes in registers, mix-down); the per-sample module arithmetic in modules.hip.h.
//
// HBM layout (all voice-minor so that lane == voice gives 256 B contiguous per wave access):
//   table   u32 [n_rows][V]           state rows, then per-voice parameter rows
//   frames  f32 [planes][T][V]
//   rings   f32 [n_rings][B][V]
//   mixpart f32 [planes][n_waves][T]  per-wave partial sums, T contiguous
//   tracks  f32 [n_tracks][T]         what the control program hands to the voice program (and its units to each other)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "fused.hip.h"
#include "interp.hip.h"
#include "jit.hpp"
#include "kernel_args.hip.h"
#include "modules.hip.h"
#include "runtime.hpp"

namespace srack {

// ====================================================================================================
// host side
// ====================================================================================================

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
            return SRACK_ERR_DEVICE;                                                                    \
        }                                                                                               \
    } while (0)

// Tuning knobs of tools/ (not part of the interface): read once per process and clamped to values the schedule can run with.
struct Knobs {
    uint32_t want_waves = 1024;  // waves to aim for when there are few voices (one per SIMD)
    uint32_t chunk_max = 4096;   // samples per launch; measured on the headline workload: 4096 -> 13.98, 8192 -> 14.14, 16384 -> 14.8 ms per step
    uint32_t chunk_first = 1024; // samples of the first chunks of a render with a control program (the control work for them is exposed)
    bool debug_occ = false;
    int high_prio_ctl = 1;       // the control stream is created with the highest priority (its own pool of hardware queues)
    int special_ctl = 1;         // specialised kernels carry the control program's units (0: control program on its own stream)
    int fm_split = 1;            // exact mode: the z^-1 FM pair on two waves per 64 voices (modulators / carriers); 0: one wave does both
    int fm_block = 1;            // default mode, buffer_size 256 ... 1024: the time-parallel FM pair with its ring in LDS; 0: render_fm_pair_ring
    uint32_t fm_block_chunk = 65536;  // ... its launch length: the ring is loaded and stored once per launch, so the whole segment by default
    uint32_t fm_block_min = 4096;     // ... and the shortest call it takes: a host that ticks block by block (one call per buffer_size samples) would move
                                      // the ring between HBM and LDS every call — config 4 at buffer_size 1024, 47 calls of 1024: 18.2 ms against 7.5 in one
                                      // call and 10.5 through render_fm_pair_ring; calls of 4096: 9.9 — so short calls keep the ring kernel
    int mix_aside = 0;                // 1: the mix partials of chunk k are summed on a side stream while chunk k + 1 renders.  Measured (round 4, one box,
                                      // three alternating rounds): config 3 11.52 / 11.54 / 11.57 ms per step all at once at the end, 11.62 / 11.67 / 11.69 aside;
                                      // cfg3_poly 16.40 against 16.43 — at the power cap a kernel that runs beside the voices is paid for in clock
    int tick = 2;                     // calls keep the control program running ahead across calls (TickSession below); 1: calls of one chunk only (round 3); 0: every call starts it afresh
};
static const Knobs& knobs()
{
    static const Knobs k = [] {
        Knobs v;
        auto num = [](const char* name, long lo, long hi, long dflt) {
            const char* e = getenv(name);
            if (!e || !*e) return dflt;
            char* end = nullptr;
            const long x = strtol(e, &end, 10);
            if (end == e) return dflt;
            return std::min(std::max(x, lo), hi);
        };
        v.want_waves = (uint32_t)num("SRACK_WANT_WAVES", 1, 1 << 20, 1024);
        v.chunk_max = (uint32_t)num("SRACK_CHUNK_MAX", 256, 65536, 4096);
        v.chunk_first = (uint32_t)num("SRACK_CHUNK_FIRST", 32, 4096, 1024);
        v.debug_occ = getenv("SRACK_DEBUG_OCC") != nullptr;
        v.high_prio_ctl = (int)num("SRACK_CTL_HIGH_PRIO", 0, 1, 1);
        v.special_ctl = (int)num("SRACK_SPECIAL_CTL", 0, 1, 1);
        v.fm_split = (int)num("SRACK_FM_SPLIT", 0, 1, 1);
        v.fm_block = (int)num("SRACK_FM_BLOCK", 0, 1, 1);
        v.fm_block_chunk = (uint32_t)num("SRACK_FM_BLOCK_CHUNK", 256, 65536, 65536);
        v.fm_block_min = (uint32_t)num("SRACK_FM_BLOCK_MIN", 1, 65536, 4096);
        v.tick = (int)num("SRACK_TICK", 0, 2, 2);
        v.mix_aside = (int)num("SRACK_MIX_ASIDE", 0, 1, 0);
        return v;
    }();
    return k;
}

struct DevProg {  // device copy of one FlatProgram
    DevOp* d_ops = nullptr;
    uint32_t* d_table = nullptr;
    float* d_rings = nullptr;
    uint32_t* d_seqtab = nullptr;
    double* d_fv = nullptr;  // OP_FREEVERB blocks: [fv_rows][n_voices] f64
    void release()
    {
        (void)hipFree(d_fv);
        d_fv = nullptr;
        (void)hipFree(d_ops);
        (void)hipFree(d_table);
        (void)hipFree(d_rings);
        (void)hipFree(d_seqtab);
        d_seqtab = nullptr;
        d_ops = nullptr;
        d_table = nullptr;
        d_rings = nullptr;
    }
};

// A host that ticks block by block — one srack_render call per buffer_size samples, the reference's own loop (main.rs:59-63) — renders
// one chunk per call.  Treated as separate renders, every call would first wait for its chunk's control tracks: the control program's
// units are one-wave latency chains, hidden only behind the voice kernels of EARLIER chunks (P1, 1024-sample calls: 14.3 ms per
// second of audio against 11.1 in one call; the sequencer patch P3: 28.6 against 18.0).  A tick session keeps the control program
// running ahead ACROSS calls: the voice launch of call c carries, as its first blocks, the units' work on the chunks of the calls to
// come (unit s, `lag[s]` launches behind the first unit, is 1 + max_lag - lag[s] chunks ahead of the voices), on the guess that the
// next call continues this one — same program, same length, same stream.  What is computed ahead must not touch what the patch holds
// as of the last rendered sample (a host may read state back, edit the patch, or ask for another length next): the units' state
// moves through a ring of R = max_lag + 2 copies of their tables (chunk x reads copy x % R — the table itself for x = 0 — and writes
// copy (x + 1) % R), the tracks through a ring of R chunk buffers.  Ending a session (another length, an edit, a state read-back)
// copies the state as of the last rendered chunk into the tables and forgets the rest.  Sample for sample the units do what they
// would do in separate calls of the same length, so a session changes no bit of any render.
// Since round 4 a call may be SEVERAL chunks (a host that renders a second of audio per call, back to back: bench.py's steps): the session's
// unit is the chunk — call j is chunks [j n, (j + 1) n) of one schedule, full chunks of Lmax samples and a shorter last one — so the
// launches of a call's last chunks already carry the control work of the next call's first ones: no exposed control launch at the start of
// a call, no ramp of short first chunks (config 2, whose five control units were five exposed launches per call: 3.44 -> see NOTES R4.8).
struct TickSession {
    bool on = false;
    uint32_t L = 0;          // samples per call
    uint32_t Lmax = 0;       // samples per full chunk (a slot of the track ring)
    uint32_t S = 0;          // floats between two tracks of the ring — and between two rows of the mix partials, which share KernelArgs::t_stride: max(L, R * Lmax)
    uint32_t n = 0;          // chunks per call
    uint32_t R = 0;          // depth of the rings
    uint64_t c = 0;          // chunks rendered (a multiple of n between calls)
    uint64_t n0 = 0;         // absolute index of the session's first sample
    hipStream_t st = nullptr;          // the stream the session's calls come on: COMPARED with the next call's, never used after the call it came with
    hipEvent_t done = nullptr;         // recorded on that stream at the end of every call of the session: whoever ends the session waits for it
    float* d_ring = nullptr;           // [n_tracks][S] control tracks: chunk x in floats [(x % R) Lmax, ... + its length) of every track
    size_t ring_bytes = 0;
    std::vector<uint32_t*> d_copies;   // per unit: [R][words] copies of its table
    std::vector<size_t> words;
    uint64_t slots_base = 0;           // argument blocks of the carried launches of calls [slots_base, slots_base + slots_n) are on the device
    uint32_t slots_n = 0;
};
constexpr uint32_t kTickBatch = 256, kTickFirstBatch = 8;  // (a session that the very next call ends should not have uploaded much)

struct DeviceState {
    TickSession tick;
    // The per-wave mix partials of chunk k are summed (mix_reduce_groups) on a side stream while chunk k + 1 renders, instead of all
    // at once after the last chunk: the 0.17 ms that pass over V/64 x T floats were the one stretch of a step without a voice kernel.
    hipStream_t mix_stream = nullptr;
    std::vector<hipEvent_t> ev_mix;   // [chunk] recorded on the render's stream after the chunk's voice launch
    hipEvent_t ev_mix_done = nullptr; // recorded on the side stream after its last sum
    DevProg voice;
    std::vector<DevProg> ctl;  // one per control stage
    KernelArgs* d_stage_slots = nullptr;  // [launches][stages] argument blocks of the staged control pipeline
    size_t stage_slots_cap = 0;
    std::vector<KernelArgs> h_stage_slots;
    float* d_mixpart = nullptr;
    size_t mixpart_bytes = 0;
    float* d_mixgroup = nullptr;
    size_t mixgroup_bytes = 0;
    float* d_tracks = nullptr;
    size_t tracks_bytes = 0;
    std::shared_ptr<const JitKernel> jit[5];  // the specialised voice kernel per output mode (1 frames, 2 mix, 3 both, 4 neither), co-owned with the cache
    bool jit_failed = false;  // specialisation was tried by default and is not available: the interpreter renders
    std::string jit_note;     // for srack_render_info: how the kernel was come by (compiled in N ms / disk cache / memory), or why there is none
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timings;  // (start, stop) pairs of the dominant kernel, not yet read
    std::vector<hipEvent_t> pool;
    const char* kernel_name = "";
    hipStream_t ctl_stream = nullptr;      // the control program runs ahead of the voice kernels on its own stream
    hipEvent_t ev_begin = nullptr;         // render start on the caller's stream
    std::vector<hipEvent_t> ev_chunk;      // control chunk k finished
    hipEvent_t ev_ready = nullptr;         // recorded on the null stream behind the upload's fills; the first render's stream waits for it (ready_pending)
    bool ready_pending = false;
};

void device_release(DeviceState* d)
{
    if (!d) return;
    d->voice.release();
    for (DevProg& c : d->ctl) c.release();
    (void)hipFree(d->d_stage_slots);
    (void)hipFree(d->d_mixpart);
    (void)hipFree(d->d_mixgroup);
    (void)hipFree(d->d_tracks);
    if (d->tick.done) (void)hipEventDestroy(d->tick.done);
    for (hipEvent_t e : d->ev_mix) (void)hipEventDestroy(e);
    if (d->ev_mix_done) (void)hipEventDestroy(d->ev_mix_done);
    if (d->mix_stream) (void)hipStreamDestroy(d->mix_stream);
    (void)hipFree(d->tick.d_ring);
    for (uint32_t* c : d->tick.d_copies) (void)hipFree(c);
    for (auto& p : d->timings) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : d->pool) (void)hipEventDestroy(e);
    for (auto e : d->ev_chunk) (void)hipEventDestroy(e);
    if (d->ev_begin) (void)hipEventDestroy(d->ev_begin);
    if (d->ev_ready) (void)hipEventDestroy(d->ev_ready);
    if (d->ctl_stream) (void)hipStreamDestroy(d->ctl_stream);
    delete d;
}

PatchHandle::~PatchHandle()
{
    device_release(dev);
    device_release(dev_old);
}

// SPECIALIZE / NO_SPECIALIZE choose a kernel for the SAME flattened program: they are not part of the program's identity.  (A host that
// reserves with flags 0 and renders with SPECIALIZE, or toggles the bit between renders, must not re-flatten — that would restart every
// voice and throw the reserved scratch and the compiled kernel away.)
constexpr uint32_t kLaunchPolicyFlags = SRACK_RENDER_SPECIALIZE | SRACK_RENDER_NO_SPECIALIZE;

static bool program_current(const PatchHandle& h, uint32_t flags)
{
    return h.prog_valid && h.prog_graph_revision == h.graph.revision && h.prog_voices_revision == h.voices_revision && h.prog_flags == (flags & ~kLaunchPolicyFlags);
}

// The program `flags` would render, WITHOUT touching the handle: the handle's own when it is current, else a flatten of a copy of the
// graph into `scratch` (srack_render_kernel_source / _compile are diagnostics: they must not restart the voices of a running patch).
int peek_program(PatchHandle& h, uint32_t flags, FlatPair& scratch, const FlatPair** out)
{
    if (program_current(h, flags)) {
        *out = &h.prog;
        return SRACK_OK;
    }
    Graph g = h.graph;
    const int rc = flatten(g, h.n_voices, h.overrides, (flags & ~kLaunchPolicyFlags) | (h.keep_state ? kFlattenEvalAll : 0u), scratch);
    *out = &scratch;
    return rc;
}

// The state as of the last rendered sample goes back into the units' tables; what was computed ahead is dropped.  The copies are issued
// on `on` — the stream of the call that ends the session, or the null stream followed by a host-side wait when no call is at hand (an
// edit, a state read-back) — after that stream has waited for the session's last call (TickSession::done).  The stream the session's
// calls came on is not touched: a host may have destroyed it since (it moved to another stream, or is tearing down).  The session
// is only forgotten once every copy has been enqueued: a failure leaves it in place and is reported.
static int tick_end(PatchHandle& h, hipStream_t on, bool host_wait)
{
    DeviceState* d = h.dev;
    if (!d || !d->tick.on) return SRACK_OK;
    TickSession& k = d->tick;
    if (k.done) HIP_TRY(hipStreamWaitEvent(on, k.done, 0));
    for (size_t s2 = 0; s2 < k.d_copies.size(); s2++)
        if (k.words[s2] > 0)  // chunk c - 1 left its state in copy c % R
            HIP_TRY(hipMemcpyAsync(d->ctl[s2].d_table, k.d_copies[s2] + (size_t)(k.c % k.R) * k.words[s2], sizeof(uint32_t) * k.words[s2], hipMemcpyDeviceToDevice, on));
    k.on = false;
    if (host_wait) HIP_TRY(hipStreamSynchronize(on));  // (whatever stream the next call comes on finds the tables in place)
    return SRACK_OK;
}

int ensure_program(PatchHandle& h, uint32_t flags)
{
    if (program_current(h, flags)) return SRACK_OK;
    {  // the program is about to be replaced: what it holds is read back (keep_state) or dropped — as of the last rendered sample either way
        const int rc_tick = tick_end(h, nullptr, true);
        if (rc_tick != SRACK_OK) return rc_tick;
    }
    flags &= ~kLaunchPolicyFlags;
    // srack_patch_keep_state: what the modules hold on the device becomes the state the re-flattened program starts from —
    // per voice for the modules of the voice program (a per-voice override of the state field), once for a module the control
    // program evaluates (it stays voice-invariant: the field itself).  Rings and reverb lines are not carried.
    const bool carry = h.keep_state && h.prog_valid && !h.voices_fresh && h.samples_rendered > 0;
    // The carried values are collected first and committed only once flatten() has succeeded: a failed flatten leaves the patch as
    // the host last saw it.  A state field the host wrote since the last flatten keeps the host's value.
    Graph carried_graph;
    std::vector<VoiceOverride> carried_ov;
    if (carry && h.dev) {
        carried_graph = h.graph;
        carried_ov = h.overrides;
        std::vector<double> values;
        for (int m = 0; m < (int)carried_graph.modules.size(); m++) {
            Module& mod = carried_graph.modules[(size_t)m];
            for (int f = 0; f < (int)mod.fields.size(); f++) {
                if (!Graph::field_is_state(mod.type, f) || h.state_writes.count({m, f}) || !read_device_state(h, m, f, values)) continue;
                const bool in_ctl = h.prog.n_tracks > 0 && m < (int)h.prog.ctl_stage.size() && h.prog.ctl_stage[(size_t)m] >= 0;
                for (auto it = carried_ov.begin(); it != carried_ov.end();) it = (it->module == m && it->field == f) ? carried_ov.erase(it) : it + 1;
                if (in_ctl)
                    mod.fields[(size_t)f] = values[0];
                else
                    carried_ov.push_back(VoiceOverride{m, f, values});
                // a sample player that has run has consumed its `wavebox.new` (sample.rs:199-203) — unless the wave was set after
                // the program that ran was flattened
                if (mod.type == SRACK_MOD_SAMPLE && mod.wave_revision <= h.prog_graph_revision) mod.fields[SRACK_SAMPLE_WAVE_NEW] = 0.0;
            }
        }
    }
    std::vector<PatchHandle::OldTag> pending_tags;  // what the program about to be replaced holds besides module fields
    if (carry && h.dev) {
        for (const auto& t : h.prog.voice.carry) pending_tags.push_back({-1, h.prog.voice.n_voices, t});
        for (size_t s = 0; s < h.prog.ctl.size(); s++)
            for (const auto& t : h.prog.ctl[s].carry) pending_tags.push_back({(int)s, h.prog.ctl[s].n_voices, t});
    }
    const bool use_carried = carry && h.dev;
    FlatPair fresh;
    int rc = flatten(use_carried ? carried_graph : h.graph, h.n_voices, use_carried ? carried_ov : h.overrides, flags | (h.keep_state ? kFlattenEvalAll : 0u), fresh);
    if (rc != SRACK_OK) return rc;
    if (use_carried) {  // commit: the carried state is now what the patch holds (the plan made by flatten travels with the graph)
        h.graph = std::move(carried_graph);
        h.overrides = std::move(carried_ov);
    }
    h.prog = std::move(fresh);
    h.state_writes.clear();
    h.prog_valid = true;
    h.prog_graph_revision = h.graph.revision;
    h.prog_voices_revision = h.voices_revision;
    h.prog_flags = flags;
    if (!carry) h.samples_rendered = 0;
    h.voices_fresh = false;
    if (h.dev) {  // device copies are rebuilt lazily by device_render
        if (carry) {  // (a second edit before any render finds h.dev empty and leaves the stash of the first in place)
            device_release(h.dev_old);
            h.dev_old = h.dev;
            h.old_tags = std::move(pending_tags);
        } else {
            device_release(h.dev);
        }
        h.dev = nullptr;
    }
    if (!carry) {
        device_release(h.dev_old);
        h.dev_old = nullptr;
        h.old_tags.clear();
    }
    return SRACK_OK;
}

// rings[row][v] = init[row] for every voice (row = ring * B + sample)
__global__ void ring_fill(float* rings, const float* init, uint32_t V)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) rings[(size_t)blockIdx.y * V + v] = init[blockIdx.y];
}

static int upload_one(const FlatProgram& P, DevProg& d)
{
    if (!P.ops.empty()) {
        HIP_TRY(hipMalloc(&d.d_ops, sizeof(DevOp) * P.ops.size()));
        HIP_TRY(hipMemcpy(d.d_ops, P.ops.data(), sizeof(DevOp) * P.ops.size(), hipMemcpyHostToDevice));
    }
    if (!P.table.empty()) {
        HIP_TRY(hipMalloc(&d.d_table, sizeof(uint32_t) * P.table.size()));
        HIP_TRY(hipMemcpy(d.d_table, P.table.data(), sizeof(uint32_t) * P.table.size(), hipMemcpyHostToDevice));
    }
    if (!P.seqtab.empty()) {
        HIP_TRY(hipMalloc(&d.d_seqtab, sizeof(uint32_t) * P.seqtab.size()));
        HIP_TRY(hipMemcpy(d.d_seqtab, P.seqtab.data(), sizeof(uint32_t) * P.seqtab.size(), hipMemcpyHostToDevice));
    }
    if (P.fv_rows > 0) {  // delay lines and filter states of the reverbs: DelayLine::new is vec![0.0; n], filter_state 0.0
        const size_t bytes = sizeof(double) * (size_t)P.fv_rows * P.n_voices;
        HIP_TRY(hipMalloc(&d.d_fv, bytes));
        HIP_TRY(hipMemset(d.d_fv, 0, bytes));
    }
    if (P.hdr.n_rings > 0) {
        size_t bytes = sizeof(float) * (size_t)P.hdr.n_rings * (size_t)P.hdr.buffer_size * P.n_voices;
        HIP_TRY(hipMalloc(&d.d_rings, bytes));
        if (P.ring_init.empty()) {
            HIP_TRY(hipMemset(d.d_rings, 0, bytes));  // AudioBuffer::new fills 0.0 (synth.rs:31-33)
        } else {  // a loaded patch: every voice starts from the saved block
            float* d_init = nullptr;
            HIP_TRY(hipMalloc(&d_init, sizeof(float) * P.ring_init.size()));
            HIP_TRY(hipMemcpy(d_init, P.ring_init.data(), sizeof(float) * P.ring_init.size(), hipMemcpyHostToDevice));
            const uint32_t n_rows = (uint32_t)P.ring_init.size();
            hipLaunchKernelGGL(ring_fill, dim3((P.n_voices + 255) / 256, n_rows), dim3(256), 0, 0, d.d_rings, d_init, P.n_voices);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(d_init));
        }
    }
    return SRACK_OK;
}

// rows of `elem`-byte items: dst[row][v] = src[row][0] for every voice (a ring or reverb that was voice-invariant becomes per-voice)
template <class T>
__global__ void rows_broadcast(T* dst, const T* src, uint32_t V)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) dst[(size_t)blockIdx.y * V + v] = src[blockIdx.y];
}

// srack_patch_keep_state, second half: feedback rings and reverb lines of the replaced program move device to device into the
// same ring / the same reverb of the new one (matched by the module and port they belong to; same length, same voice count).
static int transplant(PatchHandle& h)
{
    DeviceState* o = h.dev_old;
    HIP_TRY(hipDeviceSynchronize());  // the last render may still be running on the caller's stream
    auto prog_of = [&](DeviceState* d, int stage) -> DevProg& { return stage < 0 ? d->voice : d->ctl[(size_t)stage]; };
    auto visit = [&](int stage, const FlatProgram& P) -> int {
        for (const auto& t : P.carry)
            for (const auto& ot : h.old_tags) {
                if (ot.tag.module != t.module || ot.tag.port != t.port || ot.tag.where != t.where || ot.tag.count != t.count) continue;
                if (ot.stage >= (int)o->ctl.size()) continue;
                const DevProg &src = prog_of(o, ot.stage), &dst = prog_of(h.dev, stage);
                const size_t V = P.n_voices;
                if (ot.n_voices != P.n_voices) {
                    // the module changed sides between the per-voice program and the voice-invariant control program (V = 1)
                    const size_t Vo = ot.n_voices, rows = (size_t)t.count;
                    const size_t df = t.where == 1 ? (size_t)t.first * rows : (size_t)t.first, sf = ot.tag.where == 1 ? (size_t)ot.tag.first * rows : (size_t)ot.tag.first;
                    if (Vo != 1 && V != 1) continue;
                    if (t.where == 2) {
                        if (!src.d_fv || !dst.d_fv) continue;
                        if (Vo == 1)
                            hipLaunchKernelGGL(rows_broadcast<double>, dim3((uint32_t)((V + 255) / 256), (uint32_t)rows), dim3(256), 0, 0, dst.d_fv + df * V, src.d_fv + sf, (uint32_t)V);
                        else  // one voice stands for all: voice 0
                            HIP_TRY(hipMemcpy2D(dst.d_fv + df, sizeof(double), src.d_fv + sf * Vo, sizeof(double) * Vo, sizeof(double), rows, hipMemcpyDeviceToDevice));
                    } else {
                        uint32_t* d4 = t.where == 0 ? dst.d_table : (uint32_t*)dst.d_rings;
                        const uint32_t* s4 = t.where == 0 ? src.d_table : (const uint32_t*)src.d_rings;
                        if (!d4 || !s4) continue;
                        if (Vo == 1)
                            hipLaunchKernelGGL(rows_broadcast<uint32_t>, dim3((uint32_t)((V + 255) / 256), (uint32_t)rows), dim3(256), 0, 0, d4 + df * V, s4 + sf, (uint32_t)V);
                        else
                            HIP_TRY(hipMemcpy2D(d4 + df, 4, s4 + sf * Vo, 4 * Vo, 4, rows, hipMemcpyDeviceToDevice));
                    }
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipDeviceSynchronize());
                    break;
                }
                if (t.where == 0 && src.d_table && dst.d_table)
                    HIP_TRY(hipMemcpy(dst.d_table + (size_t)t.first * V, src.d_table + (size_t)ot.tag.first * V, sizeof(uint32_t) * (size_t)t.count * V, hipMemcpyDeviceToDevice));
                if (t.where == 1 && src.d_rings && dst.d_rings)
                    HIP_TRY(hipMemcpy(dst.d_rings + (size_t)t.first * t.count * V, src.d_rings + (size_t)ot.tag.first * t.count * V, sizeof(float) * (size_t)t.count * V, hipMemcpyDeviceToDevice));
                if (t.where == 2 && src.d_fv && dst.d_fv)
                    HIP_TRY(hipMemcpy(dst.d_fv + (size_t)t.first * V, src.d_fv + (size_t)ot.tag.first * V, sizeof(double) * (size_t)t.count * V, hipMemcpyDeviceToDevice));
                break;
            }
        return SRACK_OK;
    };
    int rc = visit(-1, h.prog.voice);
    for (size_t s = 0; s < h.prog.ctl.size() && rc == SRACK_OK; s++) rc = visit((int)s, h.prog.ctl[s]);
    return rc;
}

static int upload_program(PatchHandle& h)
{
    h.dev = new DeviceState();
    int rc = upload_one(h.prog.voice, h.dev->voice);
    h.dev->ctl.resize(h.prog.ctl.size());
    for (size_t s = 0; s < h.prog.ctl.size() && rc == SRACK_OK; s++) rc = upload_one(h.prog.ctl[s], h.dev->ctl[s]);
    // The fills above (hipMemset of rings and reverb lines) are asynchronous on the null stream, and what reads them first may run on the
    // caller's stream or on the control pipeline's non-blocking one.  Ordered by an EVENT — recorded here behind the fills, waited for by the
    // first render's stream (Segment::run; the control and mix streams start behind events of that stream) — not by putting the whole device
    // to rest: a host that edits one patch's parameters while its other patches render on other streams re-flattens without stalling them
    // (until round 6 this was a hipDeviceSynchronize).
    if (rc == SRACK_OK) {
        HIP_TRY(hipEventCreateWithFlags(&h.dev->ev_ready, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(h.dev->ev_ready, nullptr));
        h.dev->ready_pending = true;
    }
    if (rc == SRACK_OK && h.dev_old) {
        rc = transplant(h);
        device_release(h.dev_old);
        h.dev_old = nullptr;
        h.old_tags.clear();
    }
    return rc;
}

#ifndef SRK_POISON
#define SRK_POISON 0
#endif
static int grow(float*& p, size_t& have, size_t need)
{
    if (need <= have) return SRACK_OK;
    (void)hipFree(p);
    p = nullptr;
    have = 0;
    HIP_TRY(hipMalloc(&p, need));
#if SRK_POISON  // (debug build, tools/: scratch buffers start as NaNs — whatever reads a word nobody wrote shows in the output)
    HIP_TRY(hipMemset(p, 0xff, need));
#endif
    have = need;
    return SRACK_OK;
}

// ---- fused-kernel dispatch (template parameters from runtime port flags) ---------------------------
template <uint32_t A, uint32_t F, bool E, int O>
static void launch_fused4(bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (track)
        hipLaunchKernelGGL((render_voice_chain_track<A, F, E, O>), grid, dim3(64), 0, st, ka, roles, co);
    else
        hipLaunchKernelGGL((render_voice_chain<A, OSC_OUT_SQUARE, F, E, O>), grid, dim3(64), 0, st, ka, roles);
}

template <uint32_t A, uint32_t F>
static void launch_fused3(bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (exact && out_mode == 3 && track)  // frames + mix, the usual request, also gets the compile-time output mode in exact mode
        hipLaunchKernelGGL((render_voice_chain_track<A, F, true, 3>), grid, dim3(64), 0, st, ka, roles, co);
    else if (exact)  // otherwise one instantiation, output mode decided at run time
        launch_fused4<A, F, true, 0>(track, ka, roles, co, grid, st);
    else if (out_mode == 3)
        launch_fused4<A, F, false, 3>(track, ka, roles, co, grid, st);
    else if (out_mode == 1)
        launch_fused4<A, F, false, 1>(track, ka, roles, co, grid, st);
    else if (out_mode == 2)
        launch_fused4<A, F, false, 2>(track, ka, roles, co, grid, st);
    else  // neither frames nor mix requested: the render only advances the voice state
        launch_fused4<A, F, false, 0>(track, ka, roles, co, grid, st);
}

template <uint32_t A>
static void launch_fused2(uint32_t vcf_port, bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid,
                          hipStream_t st)
{
    if (vcf_port == VCF_OUT_LP)
        launch_fused3<A, VCF_OUT_LP>(exact, out_mode, track, ka, roles, co, grid, st);
    else if (vcf_port == VCF_OUT_BP)
        launch_fused3<A, VCF_OUT_BP>(exact, out_mode, track, ka, roles, co, grid, st);
    else
        launch_fused3<A, VCF_OUT_HP>(exact, out_mode, track, ka, roles, co, grid, st);
}

static void launch_fused(uint32_t osc_port, uint32_t vcf_port, bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles,
                         const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (osc_port == OSC_OUT_SAW)
        launch_fused2<OSC_OUT_SAW>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
    else if (osc_port == OSC_OUT_SQUARE)
        launch_fused2<OSC_OUT_SQUARE>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
    else
        launch_fused2<OSC_OUT_SINE>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
}

template <uint32_t kPort>
static void launch_seq2(int out_mode, const KernelArgs& ka, const SeqRoles& r, dim3 grid, hipStream_t st)
{
    if (out_mode == 3)
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 3>), grid, dim3(64), 0, st, ka, r);
    else if (out_mode == 1)
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 1>), grid, dim3(64), 0, st, ka, r);
    else
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 0>), grid, dim3(64), 0, st, ka, r);
}

static void launch_seq(uint32_t osc_port, int out_mode, const KernelArgs& ka, const SeqRoles& r, dim3 grid, hipStream_t st)
{
    if (osc_port == OSC_OUT_SAW)
        launch_seq2<OSC_OUT_SAW>(out_mode, ka, r, grid, st);
    else if (osc_port == OSC_OUT_SQUARE)
        launch_seq2<OSC_OUT_SQUARE>(out_mode, ka, r, grid, st);
    else
        launch_seq2<OSC_OUT_SINE>(out_mode, ka, r, grid, st);
}

// kRing: the feedback delay is a ring in HBM (buffer_size >= 32) instead of last tick's value in a register (buffer_size 1)
template <bool kRing>
static void launch_fm_pair2(bool exact, int out_mode, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
#define SRK_FM(E, O)                                                                                   \
    do {                                                                                               \
        if (kRing)                                                                                     \
            hipLaunchKernelGGL((render_fm_pair_ring<E, O>), grid, dim3(64), 0, st, ka, roles);         \
        else if ((E) && knobs().fm_split) /* exact mode only: measured, see the kernel's comment */    \
            hipLaunchKernelGGL((render_fm_pair_split<true, 0>), grid, dim3(128), 0, st, ka, roles);    \
        else                                                                                           \
            hipLaunchKernelGGL((render_fm_pair<E, O>), grid, dim3(64), 0, st, ka, roles);              \
    } while (0)
    if (exact)
        SRK_FM(true, 0);
    else if (out_mode == 3)
        SRK_FM(false, 3);
    else if (out_mode == 1)
        SRK_FM(false, 1);
    else if (out_mode == 2)
        SRK_FM(false, 2);
    else
        SRK_FM(false, 0);
#undef SRK_FM
}

// The FM pair with a delay of 256 ... 1024 samples in default mode: time-parallel, 32 voices per 512-thread workgroup, ring in LDS.
static bool fm_block_shape(const FlatProgram& P, uint32_t flags, uint32_t n_samples)
{
    return n_samples >= knobs().fm_block_min && P.fused == FUSED_FM_PAIR && P.fused_variant == 1 && !(flags & (SRACK_RENDER_EXACT_OSC | SRACK_RENDER_NO_FUSION)) && knobs().fm_block &&
           P.hdr.buffer_size >= kBlkChunk && P.hdr.buffer_size <= 1024;
}

static int launch_fm_block(int out_mode, const KernelArgs& ka, const ChainRoles& roles, hipStream_t st)
{
    const size_t lds = sizeof(float) * (size_t)ka.prog.buffer_size * kBlkVoices + sizeof(double) * 2 * kBlkSlices * kBlkVoices + 16;
    const dim3 grid(ka.n_waves), block(kBlkVoices * kBlkSlices);
#define SRK_BLK(O)                                                                                                              \
    do {                                                                                                                        \
        /* more than 64 KB of dynamic LDS has to be asked for — per device (the attribute belongs to the current device's copy of the \
           function) and from any thread: asked for before every launch, which costs nothing next to one */                    \
        HIP_TRY(hipFuncSetAttribute((const void*)render_fm_pair_block<O>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((render_fm_pair_block<O>), grid, block, lds, st, ka, roles);                                         \
    } while (0)
    if (out_mode == 3)
        SRK_BLK(3);
    else if (out_mode == 1)
        SRK_BLK(1);
    else if (out_mode == 2)
        SRK_BLK(2);
    else
        SRK_BLK(0);
#undef SRK_BLK
    return SRACK_OK;
}

static void launch_fm_pair(bool ring, bool exact, int out_mode, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
    if (ring)
        launch_fm_pair2<true>(exact, out_mode, ka, roles, grid, st);
    else
        launch_fm_pair2<false>(exact, out_mode, ka, roles, grid, st);
}

static void launch_interp(const FlatProgram& P, const KernelArgs& ka, hipStream_t st)
{
    size_t lds = ((size_t)P.hdr.n_rows + 2 + (size_t)P.hdr.n_tracks + (size_t)P.hdr.n_slots * P.hdr.tile) * 256;  // + zero, trash and track rows
    if (knobs().debug_occ) {  // tools/: what the runtime says about resident workgroups per CU for this LDS size
        int n = -1;
        hipError_t e = (P.render_flags & SRACK_RENDER_EXACT_OSC) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_interp<true>, 64, lds)
                                                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_interp<false>, 64, lds);
        fprintf(stderr, "[srack] render_interp lds=%zu B/wave occupancy=%d workgroups/CU (%s) waves=%u\n", lds, n, hipGetErrorString(e), ka.n_waves);
    }
    if (P.render_flags & SRACK_RENDER_EXACT_OSC)
        hipLaunchKernelGGL(render_interp<true>, dim3(ka.n_waves), dim3(64), lds, st, ka);
    else
        hipLaunchKernelGGL(render_interp<false>, dim3(ka.n_waves), dim3(64), lds, st, ka);
}

// ... and with the modulator exact as a whole (default mode's answer to the loop through its pitch, csrc/approx.cpp): render_fm_pair_block_x —
// unless the host asked for a specialised kernel or the general path by name
static bool fm_block_x_shape(const FlatProgram& P, uint32_t flags, uint32_t n_samples)
{
    return P.fm_pair_x == 1 && n_samples >= knobs().fm_block_min && knobs().fm_block && P.hdr.buffer_size >= 256 && P.hdr.buffer_size <= 1024 &&
           !(flags & (SRACK_RENDER_EXACT_OSC | SRACK_RENDER_NO_FUSION | SRACK_RENDER_SPECIALIZE));
}

// ... and at buffer_size 1 (the fed-back sine in a register): render_fm_pair_x
static bool fm_x_z1_shape(const FlatProgram& P, uint32_t flags)
{
    return P.fm_pair_x == 2 && !(flags & (SRACK_RENDER_EXACT_OSC | SRACK_RENDER_NO_FUSION | SRACK_RENDER_SPECIALIZE));
}
static void launch_fm_pair_x(int out_mode, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
    if (out_mode == 3)
        hipLaunchKernelGGL((render_fm_pair_x<3>), grid, dim3(128), 0, st, ka, roles);
    else if (out_mode == 1)
        hipLaunchKernelGGL((render_fm_pair_x<1>), grid, dim3(128), 0, st, ka, roles);
    else if (out_mode == 2)
        hipLaunchKernelGGL((render_fm_pair_x<2>), grid, dim3(128), 0, st, ka, roles);
    else
        hipLaunchKernelGGL((render_fm_pair_x<0>), grid, dim3(128), 0, st, ka, roles);
}

static int launch_fm_block_x(int out_mode, const KernelArgs& ka, const ChainRoles& roles, hipStream_t st)
{
    const size_t lds = fm_block_x_lds_bytes((uint32_t)ka.prog.buffer_size);
    const dim3 grid(ka.n_waves), block(kBlkVoices * kBlkSlices);
#define SRK_BLK(O)                                                                                                              \
    do {                                                                                                                        \
        HIP_TRY(hipFuncSetAttribute((const void*)render_fm_pair_block_x<O>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL((render_fm_pair_block_x<O>), grid, block, lds, st, ka, roles);                                       \
    } while (0)
    if (out_mode == 3)
        SRK_BLK(3);
    else if (out_mode == 1)
        SRK_BLK(1);
    else if (out_mode == 2)
        SRK_BLK(2);
    else
        SRK_BLK(0);
#undef SRK_BLK
    return SRACK_OK;
}

// The general path: a kernel specialised for the program (jit.cpp), or the tile interpreter.  Specialising costs a compilation
// (~1 s) the first time a program structure is seen, so by default it is reserved for renders wide enough to repay it.
constexpr uint32_t kSpecializeMinVoices = 4096;
// Which programs take a specialised kernel: those no hand-written kernel matches, and the shapes whose hand-written kernels the
// specialised ones outrun or tie on MI355X (P1 with everything per voice: 21.2 -> 18.1 ms per step; the sequencer-driven chain P3:
// 29.5 -> 24.5; since round 3 the z^-1 FM pair in default mode: the generator derives what render_fm_pair proves by hand — bounded pitch
// CVs, the wave's vote, a loop copy per class: the same 48 f64-rate instructions per voice-sample in the ISA — and config 4 runs 7.26 -
// 7.28 ms per step through it against 7.46 - 7.49 on the same box).  The flagship track kernel keeps its hand-written form (fixed-point
// phase), so do the FM pair's ring variant (per-tile votes on what the ring hands over) and its exact-mode two-wave split.
// SRACK_FM_FUSED=1 (tools/) keeps render_fm_pair for A/B runs.
static bool specializable_shape(const FlatProgram& P, uint32_t flags)
{
    if (P.fused == FUSED_FM_PAIR) {
        static const bool keep_fused = [] {
            const char* e = getenv("SRACK_FM_FUSED");
            return e && e[0] == '1';
        }();
        return P.fused_variant == 0 && !(flags & SRACK_RENDER_EXACT_OSC) && !keep_fused;
    }
    return P.fused == FUSED_NONE || P.fused == FUSED_VOICE_CHAIN || P.fused == FUSED_VOICE_CHAIN_SEQ;
}

static uint32_t lanes_per_wave(uint32_t V);

static int resolve_specialized(PatchHandle& h, uint32_t flags, uint32_t n_samples, int out_mode, const JitKernel** out, bool* with_ctl)
{
    *out = nullptr;
    *with_ctl = false;
    const FlatProgram& P = h.prog.voice;
    DeviceState* d = h.dev;
    if (!specializable_shape(P, flags) || (flags & SRACK_RENDER_NO_SPECIALIZE) || P.ops.empty() || fm_block_x_shape(P, flags, n_samples) || fm_x_z1_shape(P, flags)) return SRACK_OK;
    const bool forced = (flags & SRACK_RENDER_SPECIALIZE) != 0;
    if (!forced && (P.n_voices < kSpecializeMinVoices || d->jit_failed || !jit_supported(P))) return SRACK_OK;
    // the control program's units ride along in the same launches whenever the generator covers them all
    const bool ctl = h.prog.n_tracks > 0 && knobs().special_ctl && jit_ctl_supported(h.prog);
    if (!d->jit[out_mode]) {
        JitFetchInfo how;
        // (how many waves per SIMD this render has for the kernel: one wave per 64 voices on 1024 SIMDs)
        const uint32_t n_waves = (P.n_voices + lanes_per_wave(P.n_voices) - 1) / lanes_per_wave(P.n_voices);
        const int rc = jit_get(h.prog, out_mode, ctl, &d->jit[out_mode], &how, (int)std::min(4u, (n_waves + 1023u) / 1024u));
        if (rc != SRACK_OK) {
            if (forced) return rc;  // asked for explicitly: fail loudly
            d->jit_note = std::string(" jit=unavailable(") + last_error() + ")";
            if (d->jit_note.size() > 160) d->jit_note = d->jit_note.substr(0, 157) + "...)";
            for (char& c : d->jit_note)
                if (c == '\n') c = ' ';
            fprintf(stderr, "[srack] no specialised kernel for this program (%s): rendering through the pre-built kernels\n", last_error());
            d->jit_failed = true;
            return SRACK_OK;
        }
        char note[160];
        char budget[40] = "";
        if (how.waves > 0) std::snprintf(budget, sizeof budget, " regs=%d(budget for %d waves)", how.vgprs, how.waves);
        if (how.how == 2)
            std::snprintf(note, sizeof note, " jit=compiled(%.0f ms)%s", how.compile_ms, budget);
        else
            std::snprintf(note, sizeof note, " jit=%s%s", how.how == 1 ? "disk-cache" : "memory-cache", budget);
        d->jit_note = note;
    }
    *out = d->jit[out_mode].get();
    *with_ctl = ctl;
    return SRACK_OK;
}

static void launch_ctl(const FlatProgram& Cp, const KernelArgs& kc, hipStream_t st)
{
    if (Cp.fused == FUSED_CTL_GATE_ENV) {
        CtlWork w{kc.ops, kc.table, kc.frames + (size_t)Cp.ops[2].aux * kc.plane_stride, kc.T,
                  Cp.ops[0].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW | OSC_EXACT), nullptr, 0u};
        hipLaunchKernelGGL(render_ctl_gate_env, dim3(1), dim3(64), 0, st, w);
    } else if (Cp.fused == FUSED_FM_PAIR) {
        ChainRoles roles{};
        roles.adsr = 1;
        roles.osc_l = 2;
        roles.vca = 4;
        roles.osc_a = 5;
        roles.out = 6;
        roles.track = Cp.ops[0].aux;
        launch_fm_pair(Cp.fused_variant == 1, (Cp.render_flags & SRACK_RENDER_EXACT_OSC) != 0, 1, kc, roles, dim3(1), st);
    } else {
        launch_interp(Cp, kc, st);
    }
}

// How a render is scheduled.  Without a control program: one launch of the voice kernel.  With one: the
// render is cut into chunks; control chunk k (one wave, a latency chain) runs on its own stream and voice
// chunk k waits only for it, so all but the first control chunk hide behind voice kernels of earlier chunks.
static uint32_t lanes_per_wave(uint32_t V)
{
    const uint32_t want_waves = knobs().want_waves;
    uint32_t lanes = 64;
    while (lanes > 16 && (V + lanes - 1) / lanes * 2 <= want_waves) lanes >>= 1;
    return lanes;
}

// One segment [t_seg, t_seg + T) of a render of T_total samples (d_frames / d_mix point at the WHOLE render's buffers), in the order it is
// carried out: prepare (silence, scratch) -> plan (which kernel, how the control program overlaps, the chunk schedule, tick session or
// not) -> control (what the control program must have finished before the first voice launch; a session's bookkeeping) -> roles (the
// hand-written kernels' op roles) -> voices (one launch per chunk) -> mix (the partials' sums) -> finish.
struct Segment {
    PatchHandle& h;
    const FlatProgram& P;
    DeviceState* const d;
    TickSession& tk;
    const uint32_t V, C, T_total, t_seg, T, flags;
    float *d_frames, *d_mix;
    const hipStream_t st;
    int rc = SRACK_OK;
    // Voices per wave.  A full wave (64) is right whenever there are enough voices to give every SIMD work.
    // With few voices, half- or quarter-filled waves double / quadruple the number of waves: a VALU instruction
    // costs the same for 16 lanes as for 64, so this only pays while SIMDs would otherwise sit idle (VALU-bound
    // kernels: up to one wave per SIMD) or while waves are latency-bound (FM pair, interpreter: up to four).
    const bool fm_block;  // 32 voices per workgroup, whatever the voice count
    const bool fm_block_x;  // ... the same with the modulator exact (render_fm_pair_block_x)
    const bool fm_x_z1;     // the z^-1 pair with the modulator exact (render_fm_pair_x)
    const uint32_t lanes, n_waves;
    // plan
    bool has_ctl = false, co_ctl = false, special_ctl = false, tick = false;
    uint32_t n_stages = 0, n_tracks = 0, kChunkMax = 0, kChunkFirst = 0, max_lag = 0, n_chunks = 0, n_ctl_launch = 0;
    uint32_t stride = 0;  // KernelArgs::t_stride of the voice launches: floats between two rows of the mix partials and between two control tracks
    const JitKernel* special = nullptr;  // a kernel specialised for this program (jit.cpp), if the program takes one; special_ctl: with the control units as extra blocks of every launch
    std::vector<std::pair<uint32_t, uint32_t>> chunks;  // (t_off, len)
    // roles
    ChainRoles roles{};
    SeqRoles seq{};
    uint32_t osc_port = 0, vcf_port = 0, seq_port = 0;
    bool fused = false, track = false, seq_chain = false, fm_pair = false;
    // mix
    MixArgs m{};
    bool mix_aside = false;

    Segment(PatchHandle& h_, uint32_t T_total_, uint32_t t_seg_, uint32_t T_, float* d_frames_, float* d_mix_, uint32_t flags_, hipStream_t st_)
        : h(h_), P(h_.prog.voice), d(h_.dev), tk(h_.dev->tick), V(h_.prog.voice.n_voices), C((uint32_t)h_.prog.voice.hdr.n_channels), T_total(T_total_),
          t_seg(t_seg_), T(T_), flags(flags_), d_frames(d_frames_ ? d_frames_ + (size_t)t_seg_ * h_.prog.voice.n_voices : nullptr),
          d_mix(d_mix_ ? d_mix_ + t_seg_ : nullptr), st(st_), fm_block(fm_block_shape(h_.prog.voice, flags_, T_)), fm_block_x(fm_block_x_shape(h_.prog.voice, flags_, T_)), fm_x_z1(fm_x_z1_shape(h_.prog.voice, flags_)),
          lanes((fm_block || fm_block_x) ? (uint32_t)kBlkVoices : lanes_per_wave(h_.prog.voice.n_voices)), n_waves((h_.prog.voice.n_voices + lanes - 1) / lanes)
    {
    }

    // ---- small helpers -------------------------------------------------------------------------------------------------------------------
    CtlWork ctl_work(uint32_t t_off, uint32_t len)
    {
        const FlatProgram& Cp = h.prog.ctl[0];
        return CtlWork{d->ctl[0].d_ops, d->ctl[0].d_table, d->d_tracks + (size_t)Cp.ops[2].aux * T + t_off, len,
                       Cp.ops[0].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW | OSC_EXACT), nullptr, 0u};
    }
    int get_event(hipEvent_t& e)
    {
        if (!d->pool.empty()) {
            e = d->pool.back();
            d->pool.pop_back();
            return SRACK_OK;
        }
        HIP_TRY(hipEventCreate(&e));
        return SRACK_OK;
    }

    // One argument block per (launch, control unit): launch j runs unit s on chunk j - lag[s]; chunk c is complete after launch
    // c + max_lag.  (A control program that was not cut into units is one unit with lag 0.)
    KernelArgs stage_args(uint32_t s, uint32_t k)
    {
        const uint32_t t_off = chunks[k].first, len = chunks[k].second;
        KernelArgs kc{};
        kc.ops = d->ctl[s].d_ops;
        kc.prog = h.prog.ctl[s].hdr;
        kc.table = d->ctl[s].d_table;
        kc.rings = d->ctl[s].d_rings;
        kc.seqtab = d->ctl[s].d_seqtab;
        kc.fv = d->ctl[s].d_fv;
        kc.frames = d->d_tracks + t_off;  // the control program's planes are the tracks: [n_tracks][T][1]
        kc.tracks = d->d_tracks + t_off;  // ... and later stages read earlier stages' tracks from the same buffer
        kc.plane_stride = T;
        kc.t_stride = T;
        kc.V = 1;
        kc.T = len;
        kc.n_waves = 1;
        kc.lanes = 64;
        kc.n0 = h.samples_rendered + t_off;
        return kc;
    }
    int upload_stage_slots(hipStream_t on)
    {
        d->h_stage_slots.assign((size_t)n_ctl_launch * n_stages, KernelArgs{});
        for (uint32_t s2 = 0; s2 < n_stages; s2++)
            for (uint32_t k = 0; k < n_chunks; k++) {
                KernelArgs& slot = d->h_stage_slots[(size_t)(k + (uint32_t)h.prog.ctl_lag[s2]) * n_stages + s2];
                slot = stage_args(s2, k);
                slot.block0 = s2;  // the unit's one wave is wave 0 of its program
            }
        const size_t bytes = sizeof(KernelArgs) * d->h_stage_slots.size();
        if (bytes > d->stage_slots_cap) {
            (void)hipFree(d->d_stage_slots);
            d->d_stage_slots = nullptr;
            d->stage_slots_cap = 0;
            HIP_TRY(hipMalloc(&d->d_stage_slots, bytes));
            d->stage_slots_cap = bytes;
        }
        HIP_TRY(hipMemcpyAsync(d->d_stage_slots, d->h_stage_slots.data(), bytes, hipMemcpyHostToDevice, on));
        return SRACK_OK;
    }
    // ---- tick session: argument blocks of unit s2 on chunk x, and the launches that start a session ----
    float* tick_tracks(uint64_t x)
    {
        return tk.d_ring + (size_t)(x % tk.R) * tk.Lmax;
    }
    uint32_t tick_len(uint64_t x)  // chunk x of the session: its length ...
    {
        const uint32_t k = (uint32_t)(x % tk.n);
        return std::min(tk.Lmax, tk.L - k * tk.Lmax);
    }
    uint64_t tick_n0(uint64_t x)  // ... and the absolute index of its first sample
    {
        return tk.n0 + (x / tk.n) * tk.L + (x % tk.n) * (uint64_t)tk.Lmax;
    }
    uint32_t* tick_table(uint32_t s2, uint64_t x)
    {  // the copy chunk x reads (x = 0: the table itself) / chunk x - 1 wrote
        if (tk.words[s2] == 0) return nullptr;
        return x == 0 ? d->ctl[s2].d_table : tk.d_copies[s2] + (size_t)(x % tk.R) * tk.words[s2];
    }
    KernelArgs tick_unit_args(uint32_t s2, uint64_t x)
    {
        KernelArgs kc{};
        kc.ops = d->ctl[s2].d_ops;
        kc.prog = h.prog.ctl[s2].hdr;
        kc.table = tick_table(s2, x);
        kc.table_out = tick_table(s2, x + 1);
        kc.seqtab = d->ctl[s2].d_seqtab;
        kc.frames = tick_tracks(x);
        kc.tracks = kc.frames;
        kc.plane_stride = tk.S;
        kc.t_stride = tk.S;
        kc.V = 1;
        kc.T = tick_len(x);
        kc.n_waves = 1;
        kc.lanes = 64;
        kc.n0 = tick_n0(x);
        kc.block0 = s2;
        return kc;
    }
    CtlWork tick_work(uint64_t x)
    {  // the fused gate -> envelope control program on chunk x
        const FlatProgram& Cp = h.prog.ctl[0];
        return CtlWork{d->ctl[0].d_ops, tick_table(0, x), tick_tracks(x) + (size_t)Cp.ops[2].aux * tk.S, tick_len(x),
                       Cp.ops[0].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW | OSC_EXACT), tick_table(0, x + 1), (uint32_t)Cp.hdr.n_rows};
    }
    // global launch g of the session runs unit s2 on chunk g - lag[s2]: launches 0 .. max_lag run alone when the session starts (they
    // complete chunk 0), launch max_lag + 1 + i rides on the voice launch of chunk i.  The device holds the first ones and a batch of the others.
    int tick_upload_slots(uint64_t base)
    {
        const uint32_t fill = max_lag + 1, batch = std::max(base == 0 ? kTickFirstBatch : kTickBatch, tk.n);  // (at least a whole call's chunks)
        d->h_stage_slots.assign((size_t)(fill + batch) * n_stages, KernelArgs{});
        for (uint32_t s2 = 0; s2 < n_stages; s2++) {
            const uint32_t lag = (uint32_t)h.prog.ctl_lag[s2];
            if (base == 0)
                for (uint32_t g = lag; g < fill; g++) d->h_stage_slots[(size_t)g * n_stages + s2] = tick_unit_args(s2, g - lag);
            for (uint32_t i = 0; i < batch; i++) d->h_stage_slots[(size_t)(fill + i) * n_stages + s2] = tick_unit_args(s2, base + i + fill - lag);
        }
        const size_t bytes = sizeof(KernelArgs) * d->h_stage_slots.size();
        if (bytes > d->stage_slots_cap) {
            (void)hipFree(d->d_stage_slots);
            d->d_stage_slots = nullptr;
            d->stage_slots_cap = 0;
            HIP_TRY(hipMalloc(&d->d_stage_slots, bytes));
            d->stage_slots_cap = bytes;
        }
        HIP_TRY(hipMemcpyAsync(d->d_stage_slots, d->h_stage_slots.data(), bytes, hipMemcpyHostToDevice, st));
        tk.slots_base = base;
        tk.slots_n = batch;
        return SRACK_OK;
    }
    // ---- prepare: a program nothing of which reaches the output; the mix scratch -----------------------------------------------------------
    int prepare(bool& done)
    {
        done = false;
        if (P.hdr.n_planes == 0) {  // nothing reaches the output: silence (output.rs:55)
            if (d_mix)
                for (uint32_t c = 0; c < C; c++)
                    hipLaunchKernelGGL(fill_zero, dim3((T + 255) / 256), dim3(256), 0, st, d_mix + (size_t)c * T_total, (size_t)T);
            // Under srack_patch_keep_state the program holds every planned module (the reference's execute() ticks them all, heard or
            // not): they must keep running — phases, envelopes, sequencer steps, rings — or a wire patched into the output later would
            // find them frozen while the sample counter moved on.  The kernels run with nothing to write.
            if (!h.keep_state || (P.ops.empty() && h.prog.n_tracks == 0)) {
                h.samples_rendered += T;
                done = true;
                return SRACK_OK;
            }
            d_frames = nullptr;
            d_mix = nullptr;
        }

        return SRACK_OK;
    }

    // ---- plan: the kernel, the way the control program overlaps with the voices, tick session or not, the chunk schedule -------------------
    int plan()
    {
        has_ctl = h.prog.n_tracks > 0;
        // Two ways to overlap the control program with the voice kernels:
        //  co-scheduled (fused track kernel + fused gate-envelope control program): voice launch k carries one extra
        //    block that computes the track of chunk k+1; everything stays on the caller's stream.
        //  two streams (any other combination): control chunks run on a private stream, voice chunk k waits on event k.
        //    This overlaps only while the two streams map to different hardware queues (GPU_MAX_HW_QUEUES, default 4,
        //    shared with the host's other streams): measured 15 ms -> 20 ms per step once RCCL's streams are alive.
        n_stages = (uint32_t)h.prog.ctl.size();
        co_ctl = has_ctl && P.fused == FUSED_VOICE_CHAIN_TRACK && n_stages == 1 && h.prog.ctl[0].fused == FUSED_CTL_GATE_ENV && h.prog.n_tracks == 1;
        {
            const int om = (d_frames ? 1 : 0) | (d_mix ? 2 : 0);
            if ((rc = resolve_specialized(h, flags, T, om ? om : 4, &special, &special_ctl)) != SRACK_OK) return rc;
        }
        // chunk schedule: short first chunks (only control chunk 0 is exposed), doubling up to kChunkMax
        // With a control pipeline of depth L the first voice chunk waits for L + 1 control launches: those stay short.
        kChunkMax = knobs().chunk_max;
        kChunkFirst = std::min(knobs().chunk_first, kChunkMax);
        max_lag = 0;
        for (int lag : h.prog.ctl_lag) max_lag = std::max(max_lag, (uint32_t)lag);
        // A call of one chunk whose control program runs as blocks of the voice launches is (the start of) a tick session (TickSession):
        // everything the units hold must live in their tables (no rings in HBM, no reverb lines — those have no copies to move through).
        tick = knobs().tick && has_ctl && (co_ctl || (special && special_ctl)) && t_seg == 0 && T_total == T && (T <= kChunkMax || knobs().tick >= 2);
        for (const FlatProgram& Cp : h.prog.ctl) tick = tick && Cp.hdr.n_rings == 0 && Cp.fv_rows == 0;
        const uint32_t tick_n = (T + kChunkMax - 1) / kChunkMax;  // chunks of a ticked call: full ones and a shorter last one
        if (tk.on && !(tick && tk.L == T && tk.Lmax == std::min(T, kChunkMax) && tk.st == st && tk.c % tk.n == 0 &&
                       tk.n0 + (tk.c / tk.n) * (uint64_t)T == h.samples_rendered)) {  // not the call the session guessed
            if ((rc = tick_end(h, st, false)) != SRACK_OK) return rc;  // on THIS call's stream, behind the session's last call (an event: no host-side wait)
        } else if (tk.on && tk.done) {
            // the session continues — on a stream with the same HANDLE as its last call's.  A host may have destroyed that stream and been
            // handed the same value for a new one: order this call behind the session's last one explicitly (an event wait on a stream that
            // already is behind it costs nothing)
            HIP_TRY(hipStreamWaitEvent(st, tk.done, 0));
        }
        chunks.clear();
        if (tick) {
            for (uint32_t k = 0; k < tick_n; k++) chunks.emplace_back(k * kChunkMax, std::min(kChunkMax, T - k * kChunkMax));
        } else if (has_ctl) {
            uint32_t k = 0;
            for (uint32_t t_off = 0, len = kChunkFirst; t_off < T; t_off += len, k++) {
                if (k > max_lag) len = std::min(len * 2, kChunkMax);
                len = std::min(len, T - t_off);
                chunks.emplace_back(t_off, len);
            }
        } else {  // no control program to overlap with, but short launches still win: the waves of a launch stay within a few
                  // samples of each other, so their frame rows land in the same DRAM pages (FM pair 13.2 -> 9.9 ms per step)
            // The z^-1 FM pair (one wave per SIMD at config 4's 65 536 voices, no ring traffic) wants them shorter still: 2048 samples 7.26 ms per
            // step, 4096 7.39, 1536 7.31, 1024 7.43, 8192 7.85 (tools/ab_env.sh, one box); its ring variant and the flagship are flat from 3072 to 6144.
            // A specialised kernel without rings in HBM at one wave per SIMD or fewer is in the same position (config 4 through the general
            // path: 7.36 - 7.46 ms per step at 4096, 7.26 - 7.28 at 2048, two rounds on one box).
            const bool fm_z1 = fm_x_z1 || (P.fused == FUSED_FM_PAIR && P.fused_variant == 0 && !(flags & (SRACK_RENDER_NO_FUSION | SRACK_RENDER_EXACT_OSC)));
            const bool lone_waves = special && P.hdr.n_rings == 0 && n_waves <= 1024 && !(flags & SRACK_RENDER_EXACT_OSC);
            // (the time-parallel FM pair keeps its ring in LDS for a launch and moves it to and from HBM at the ends: one launch per segment)
            const uint32_t len = (fm_block || fm_block_x) ? knobs().fm_block_chunk : (fm_z1 || lone_waves) ? std::min(kChunkMax, 2048u) : kChunkMax;
            for (uint32_t t_off = 0; t_off < T; t_off += len) chunks.emplace_back(t_off, std::min(len, T - t_off));
        }
        n_chunks = (uint32_t)chunks.size();
        n_ctl_launch = n_chunks + max_lag;
        n_tracks = (uint32_t)h.prog.n_tracks;
        // (a ticked call reads its tracks from the session's ring, R slots of a full chunk per track: the one stride must cover both)
        stride = tick ? std::max(T, (max_lag + 2) * std::min(T, kChunkMax)) : T;
        if (d_mix && (rc = grow(d->d_mixpart, d->mixpart_bytes, sizeof(float) * (size_t)P.hdr.n_planes * n_waves * stride)) != SRACK_OK) return rc;
        if (d_mix && (rc = grow(d->d_mixgroup, d->mixgroup_bytes, sizeof(float) * (size_t)P.hdr.n_planes * kMixSplit * stride)) != SRACK_OK) return rc;
        if (has_ctl && (rc = grow(d->d_tracks, d->tracks_bytes, sizeof(float) * (size_t)h.prog.n_tracks * T)) != SRACK_OK) return rc;
        // One argument block per (launch, control unit): launch j runs unit s on chunk j - lag[s]; chunk c is complete after launch
        // c + max_lag.  (A control program that was not cut into units is one unit with lag 0.)
        return SRACK_OK;
    }

    // ---- control: what must run before the first voice launch (a session's start and bookkeeping; chunk 0's control work; the two-stream
    // fallback's whole control pipeline) ---------------------------------------------------------------------------------------------------
    int control()
    {
        if (tick && !tk.on) {  // a session starts: its first chunk's control work is exposed, like any render's
            const uint32_t R = max_lag + 2;
            if ((rc = grow(tk.d_ring, tk.ring_bytes, sizeof(float) * (size_t)n_tracks * stride)) != SRACK_OK) return rc;
            if (tk.d_copies.empty()) {
                tk.d_copies.assign(n_stages, nullptr);
                tk.words.assign(n_stages, 0);
                for (uint32_t s2 = 0; s2 < n_stages; s2++) {
                    tk.words[s2] = h.prog.ctl[s2].table.size();
                    if (tk.words[s2] > 0) {
                        HIP_TRY(hipMalloc(&tk.d_copies[s2], sizeof(uint32_t) * tk.words[s2] * R));
#if SRK_POISON
                        HIP_TRY(hipMemset(tk.d_copies[s2], 0xff, sizeof(uint32_t) * tk.words[s2] * R));
#endif
                    }
                }
            }
            tk.on = true;
            tk.L = T;
            tk.Lmax = std::min(T, kChunkMax);
            tk.S = stride;
            tk.n = n_chunks;
            tk.R = R;
            tk.c = 0;
            tk.n0 = h.samples_rendered;
            tk.st = st;
            if (co_ctl) {
                hipLaunchKernelGGL(render_ctl_gate_env, dim3(1), dim3(64), 0, st, tick_work(0));
                HIP_TRY(hipGetLastError());
            } else {
                if ((rc = tick_upload_slots(0)) != SRACK_OK) return rc;
                for (uint32_t g = 0; g <= max_lag; g++) {
                    KernelArgs kp{};
                    kp.block0 = n_stages;
                    kp.ctl_slots = d->d_stage_slots + (size_t)g * n_stages;
                    if ((rc = jit_launch(*special, kp, n_stages, st)) != SRACK_OK) return rc;
                }
            }
        } else if (tick && !co_ctl && tk.c + n_chunks > tk.slots_base + tk.slots_n) {  // this call's chunks need argument blocks the device does not hold yet
            if ((rc = tick_upload_slots(tk.c)) != SRACK_OK) return rc;
        }
        if (tick) {
            // (below: the voice launch of this call, with the units' next launch as its first blocks)
        } else if (special && special_ctl) {
            // Co-scheduled control units: everything on the caller's stream.  Launches 0 .. max_lag of the control pipeline run alone
            // (they complete chunk 0: the only exposed control work, kept short by the 1024-sample first chunks); voice launch k then
            // carries control launch k + max_lag + 1 as its first blocks, which completes chunk k + 1 while chunk k is consumed.
            if ((rc = upload_stage_slots(st)) != SRACK_OK) return rc;
            for (uint32_t j = 0; j <= max_lag && j < n_ctl_launch; j++) {
                KernelArgs kp{};
                kp.block0 = n_stages;
                kp.ctl_slots = d->d_stage_slots + (size_t)j * n_stages;
                if ((rc = jit_launch(*special, kp, n_stages, st)) != SRACK_OK) return rc;
            }
        } else if (co_ctl) {  // chunk 0's track: the only control work that is not hidden (1024 samples, ~0.15 ms)
            hipLaunchKernelGGL(render_ctl_gate_env, dim3(1), dim3(64), 0, st, ctl_work(chunks[0].first, chunks[0].second));
            HIP_TRY(hipGetLastError());
        } else if (has_ctl) {
            if (!d->ctl_stream) {
                // Highest priority: HIP keeps a separate pool of hardware queues per priority, so this stream does not end up sharing
                // a queue with the caller's stream once other libraries (RCCL) have created streams of their own — the overlap of
                // control chunks with voice chunks depends on the two running on different queues.
                int prio_lo = 0, prio_hi = 0;
                if (knobs().high_prio_ctl && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_hi != prio_lo)
                    HIP_TRY(hipStreamCreateWithPriority(&d->ctl_stream, hipStreamNonBlocking, prio_hi));
                else
                    HIP_TRY(hipStreamCreateWithFlags(&d->ctl_stream, hipStreamNonBlocking));
            }
            if (!d->ev_begin) HIP_TRY(hipEventCreateWithFlags(&d->ev_begin, hipEventDisableTiming));
            while (d->ev_chunk.size() < n_chunks) {
                hipEvent_t e;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                d->ev_chunk.push_back(e);
            }
            // the track buffer may still be read by the previous render on `st`: start after it
            HIP_TRY(hipEventRecord(d->ev_begin, st));
            HIP_TRY(hipStreamWaitEvent(d->ctl_stream, d->ev_begin, 0));
            const bool staged = h.prog.ctl[0].fused == FUSED_NONE;  // the interpreter: all stages side by side in one launch
            if (staged) {
                const uint32_t n_launch = n_ctl_launch;
                size_t lds = 0;
                for (uint32_t s2 = 0; s2 < n_stages; s2++) {
                    const DevProgram& H = h.prog.ctl[s2].hdr;
                    lds = std::max(lds, ((size_t)H.n_rows + 2 + (size_t)H.n_tracks + (size_t)H.n_slots * H.tile) * 256);
                }
                if ((rc = upload_stage_slots(d->ctl_stream)) != SRACK_OK) return rc;
                for (uint32_t j = 0; j < n_launch; j++) {
                    const KernelArgs* slots = d->d_stage_slots + (size_t)j * n_stages;
                    if (flags & SRACK_RENDER_EXACT_OSC)
                        hipLaunchKernelGGL(render_interp_stages<true>, dim3(n_stages), dim3(64), lds, d->ctl_stream, slots);
                    else
                        hipLaunchKernelGGL(render_interp_stages<false>, dim3(n_stages), dim3(64), lds, d->ctl_stream, slots);
                    HIP_TRY(hipGetLastError());
                    if (j >= max_lag) HIP_TRY(hipEventRecord(d->ev_chunk[j - max_lag], d->ctl_stream));
                }
            } else {
                for (uint32_t k = 0; k < n_chunks; k++) {
                    launch_ctl(h.prog.ctl[0], stage_args(0, k), d->ctl_stream);
                    HIP_TRY(hipGetLastError());
                    HIP_TRY(hipEventRecord(d->ev_chunk[k], d->ctl_stream));
                }
            }
        }

        return SRACK_OK;
    }

    // ---- roles: which op is what for the hand-written kernels, and the kernel's name for srack_render_info --------------------------------
    void pick_roles()
    {
        fused = P.fused == FUSED_VOICE_CHAIN || P.fused == FUSED_VOICE_CHAIN_TRACK;
        track = P.fused == FUSED_VOICE_CHAIN_TRACK;
        if (fused) {
            for (int i = 0; i < (int)P.ops.size(); i++) {
                const DevOp& op = P.ops[(size_t)i];
                if (op.kind == OP_VCF) { roles.vcf = i; vcf_port = op.flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP); }
                if (op.kind == OP_ADSR) roles.adsr = i;
                if (op.kind == OP_VCA) roles.vca = i;
                if (op.kind == OP_OUT) roles.out = i;
                if (op.kind == OP_VCA && track) roles.track = P.hdr.track_id[op.in_slot[1] - kTrackSlot];
            }
            const Graph& g = h.graph;
            roles.osc_a = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.vcf].module].in[0].src];
            if (!track) roles.osc_l = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.adsr].module].in[0].src];
            osc_port = P.ops[(size_t)roles.osc_a].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
            d->kernel_name = track ? "render_voice_chain_track" : "render_voice_chain";
        } else {
            d->kernel_name = "render_interp";
        }

        seq_chain = P.fused == FUSED_VOICE_CHAIN_SEQ;
        if (seq_chain) {
            seq.math = seq.trk_cutoff = -1;
            auto track_row = [&](int slot) { return P.hdr.track_id[slot - kTrackSlot]; };
            for (int i = 0; i < (int)P.ops.size(); i++) {
                const DevOp& op = P.ops[(size_t)i];
                if (op.kind == OP_MATH) { seq.math = i; seq.trk_pitch = track_row(op.in_slot[0]); }
                if (op.kind == OP_OSC) { seq.osc = i; seq_port = op.flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW); }
                if (op.kind == OP_VCF) { seq.vcf = i; if (op.flags & VCF_HAS_CV) seq.trk_cutoff = track_row(op.in_slot[1]); }
                if (op.kind == OP_VCA) { seq.vca = i; seq.trk_env = track_row(op.in_slot[1]); }
                if (op.kind == OP_OUT) {
                    if (op.in_slot[0] >= kTrackSlot) {
                        seq.extra_plane[seq.n_extra] = op.aux;
                        seq.extra_trk[seq.n_extra++] = track_row(op.in_slot[0]);
                    } else {
                        seq.out = i;
                    }
                }
            }
            if (seq.math < 0) seq.trk_pitch = track_row(P.ops[(size_t)seq.osc].in_slot[0]);
            d->kernel_name = "render_voice_chain_seq";
        }
        if (special) d->kernel_name = "render_specialized";
        fm_pair = P.fused == FUSED_FM_PAIR || fm_block_x || fm_x_z1;
        if (fm_pair) {  // op order fixed by the matcher: DELAY_RD, MATH_FB, OSC_M, DELAY_WR, MATH_IDX, OSC_C, OUT
            roles.adsr = 1;
            roles.osc_l = 2;
            roles.vca = 4;
            roles.osc_a = 5;
            roles.out = 6;
            roles.track = P.ops[0].aux;  // the ring's state row
            if (!special) d->kernel_name = fm_x_z1 ? "render_fm_pair_x" : fm_block_x ? "render_fm_pair_block_x" : fm_block ? "render_fm_pair_block" : P.fused_variant == 1 ? "render_fm_pair_ring" : "render_fm_pair";
        }
    }

    // ---- voices: one launch per chunk (with the control program's next work as its first blocks where that is how they overlap) ---------
    int voices()
    {
        if (d_mix) {
            m.mixpart = d->d_mixpart;
            m.mixgroup = d->d_mixgroup;
            m.mix = d_mix;
            m.T = stride;  // (the partials' row pitch; the samples summed are [t_begin, t_end))
            m.mix_stride = T_total;
            m.n_waves = n_waves;
            m.n_channels = C;
            m.n_planes = (uint32_t)P.hdr.n_planes;
            for (int c = 0; c < 8; c++) m.channel_plane[c] = P.hdr.channel_plane[c];
        }
        mix_aside = d_mix && n_chunks > 1 && knobs().mix_aside;  // chunk k's partials are summed beside chunk k + 1's voices
        if (mix_aside) {
            if (!d->mix_stream) HIP_TRY(hipStreamCreateWithFlags(&d->mix_stream, hipStreamNonBlocking));
            if (!d->ev_mix_done) HIP_TRY(hipEventCreateWithFlags(&d->ev_mix_done, hipEventDisableTiming));
            while (d->ev_mix.size() < n_chunks) {
                hipEvent_t e = nullptr;
                HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                d->ev_mix.push_back(e);
            }
        }
        for (uint32_t k = 0; k < n_chunks; k++) {
            const uint32_t t_off = chunks[k].first, len = chunks[k].second;
            KernelArgs ka{};
            ka.ops = d->voice.d_ops;
            ka.prog = P.hdr;
            ka.table = d->voice.d_table;
            ka.rings = d->voice.d_rings;
            ka.seqtab = d->voice.d_seqtab;
            ka.fv = d->voice.d_fv;
            ka.frames = d_frames ? d_frames + (size_t)t_off * V : nullptr;
            ka.mixpart = d_mix ? d->d_mixpart + t_off : nullptr;
            ka.tracks = tick ? tick_tracks(tk.c + k) : has_ctl ? d->d_tracks + t_off : nullptr;
            ka.plane_stride = (uint64_t)T_total * V;
            ka.t_stride = stride;
            ka.V = V;
            ka.T = len;
            ka.n_waves = n_waves;
            ka.lanes = lanes;
            ka.n0 = h.samples_rendered + t_off;
            CtlWork co{};
            if (tick && co_ctl) {  // block 0: the track of the next chunk — this call's, or the first of the call the session expects next
                co = tick_work(tk.c + k + 1);
                ka.block0 = 1;
            } else if (tick) {
                ka.block0 = n_stages;
                ka.ctl_slots = d->d_stage_slots + (size_t)(max_lag + 1 + (tk.c + k - tk.slots_base)) * n_stages;
            } else if (co_ctl && k + 1 < n_chunks) {  // this launch's block 0 prepares the next chunk's track
                co = ctl_work(chunks[k + 1].first, chunks[k + 1].second);
                ka.block0 = 1;
            }
            if (has_ctl && !co_ctl && !(special && special_ctl)) HIP_TRY(hipStreamWaitEvent(st, d->ev_chunk[k], 0));
            if (!tick && special && special_ctl && k + max_lag + 1 < n_ctl_launch) {  // this launch's first blocks: the control units' next launch
                ka.block0 = n_stages;
                ka.ctl_slots = d->d_stage_slots + (size_t)(k + max_lag + 1) * n_stages;
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            const bool timed = h.timing_armed;  // only a host that asked for srack_render_kernel_ms pays for the event pair
            if (timed) {
                if ((rc = get_event(e0)) != SRACK_OK || (rc = get_event(e1)) != SRACK_OK) return rc;
                HIP_TRY(hipEventRecord(e0, st));
            }
            if (special) {
                if ((rc = jit_launch(*special, ka, n_waves + ka.block0, st)) != SRACK_OK) return rc;
            } else if (fused) {
                const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
                launch_fused(osc_port, vcf_port, (flags & SRACK_RENDER_EXACT_OSC) != 0, out_mode, track, ka, roles, co, dim3(n_waves + ka.block0), st);
            } else if (seq_chain) {
                const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
                launch_seq(seq_port, out_mode, ka, seq, dim3(n_waves), st);
            } else if (fm_pair && fm_x_z1) {
                launch_fm_pair_x((ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0), ka, roles, dim3(n_waves), st);
            } else if (fm_pair && fm_block_x) {
                if ((rc = launch_fm_block_x((ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0), ka, roles, st)) != SRACK_OK) return rc;
            } else if (fm_pair && fm_block) {
                if ((rc = launch_fm_block((ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0), ka, roles, st)) != SRACK_OK) return rc;
            } else if (fm_pair) {
                const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
                launch_fm_pair(P.fused_variant == 1, (flags & SRACK_RENDER_EXACT_OSC) != 0, out_mode, ka, roles, dim3(n_waves), st);
            } else {
                launch_interp(P, ka, st);
            }
            HIP_TRY(hipGetLastError());
            if (mix_aside && k + 1 < n_chunks) {
                HIP_TRY(hipEventRecord(d->ev_mix[k], st));
                HIP_TRY(hipStreamWaitEvent(d->mix_stream, d->ev_mix[k], 0));
                MixArgs mk = m;
                mk.t_begin = t_off;
                mk.t_end = t_off + len;
                hipLaunchKernelGGL(mix_reduce_groups, dim3((len + 255) / 256, kMixSplit), dim3(256), 0, d->mix_stream, mk);
                HIP_TRY(hipGetLastError());
            }
            if (timed) {
                HIP_TRY(hipEventRecord(e1, st));
                d->timings.emplace_back(e0, e1);
                if (d->timings.size() > 4096) {  // nobody is reading them: recycle the oldest
                    d->pool.push_back(d->timings.front().first);
                    d->pool.push_back(d->timings.front().second);
                    d->timings.erase(d->timings.begin());
                }
            }
        }

        return SRACK_OK;
    }

    // ---- mix: the per-wave partials' deterministic sums -----------------------------------------------------------------------------------
    int mix()
    {
        if (d_mix) {
            // what is left to sum: everything, or — the earlier chunks being summed on the side stream — the last chunk only
            m.t_begin = mix_aside ? chunks[n_chunks - 1].first : 0u;
            m.t_end = T;
            hipLaunchKernelGGL(mix_reduce_groups, dim3((m.t_end - m.t_begin + 255) / 256, kMixSplit), dim3(256), 0, st, m);
            if (mix_aside) {
                HIP_TRY(hipEventRecord(d->ev_mix_done, d->mix_stream));
                HIP_TRY(hipStreamWaitEvent(st, d->ev_mix_done, 0));  // (also keeps the NEXT render's voice launches off the partials until they are summed)
            }
            hipLaunchKernelGGL(mix_reduce_final, dim3((T + 255) / 256), dim3(256), 0, st, m);
            HIP_TRY(hipGetLastError());
        }
        return SRACK_OK;
    }

    // ---- finish: the sample counter; a session's event and count -------------------------------------------------------------------------
    int finish()
    {
        h.samples_rendered += T;
        if (tick) {
            if (!tk.done) HIP_TRY(hipEventCreateWithFlags(&tk.done, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(tk.done, st));  // what ends the session waits for this call, on whatever stream it ends it
            tk.c += n_chunks;
        }
        return SRACK_OK;
    }

    int run()
    {
        bool done = false;
        if ((rc = prepare(done)) != SRACK_OK || done) return rc;
        // (a call that fails part-way — a launch error — must not leave a session behind whose bookkeeping is a call ahead of the device.  It
        // ends the session and drops the program: the ring of table copies cannot be trusted as a rollback — a session's first call has not
        // written copy 0 yet, and a call of several chunks overwrites the copy its own start would restore — so the next render starts from
        // what the host holds rather than from tables nobody can vouch for)
        struct TickGuard {
            PatchHandle& h;
            TickSession& t;
            bool done = false;
            ~TickGuard()
            {
                if (done || !t.on) return;
                (void)hipDeviceSynchronize();
                t.on = false;
                h.prog_valid = false;
            }
        } tick_guard{h, tk};
        if ((rc = plan()) != SRACK_OK || (rc = control()) != SRACK_OK) return rc;
        pick_roles();
        if ((rc = voices()) != SRACK_OK || (rc = mix()) != SRACK_OK || (rc = finish()) != SRACK_OK) return rc;
        tick_guard.done = true;
        return SRACK_OK;
    }
};

static int render_segment(PatchHandle& h, uint32_t T_total, uint32_t t_seg, uint32_t T, float* d_frames, float* d_mix, uint32_t flags, hipStream_t st)
{
    if (!h.dev) {
        const int rc = upload_program(h);
        if (rc != SRACK_OK) return rc;
    }
    if (h.dev->ready_pending) {  // the upload's fills (null stream) before anything of this render
        HIP_TRY(hipStreamWaitEvent(st, h.dev->ev_ready, 0));
        h.dev->ready_pending = false;
    }
    return Segment(h, T_total, t_seg, T, d_frames, d_mix, flags, st).run();
}

// A render is cut into segments of at most kSegment samples so that the scratch it needs (per-wave mix partials
// [planes][V/64][T], control tracks [n_tracks][T]) stays bounded however long the render is: 16 MB of partials per
// 1000 samples at 262 144 voices.  Voice and control state carry over between segments exactly as between calls.
int device_render(PatchHandle& h, uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags, void* stream)
{
    int rc = ensure_program(h, flags);
    if (rc != SRACK_OK) return rc;
    if (n_samples == 0) return SRACK_OK;
    if (!h.dev) {
        rc = upload_program(h);
        if (rc != SRACK_OK) return rc;
    }
    // e.g. the exact oscillator forced for a patch whose approximated ports drive a pitch; the kernel choice stays this call's
    flags = h.prog.effective_flags | (flags & kLaunchPolicyFlags);
    constexpr uint32_t kSegment = 65536;
    for (uint32_t t = 0; t < n_samples && rc == SRACK_OK; t += kSegment)
        rc = render_segment(h, n_samples, t, std::min(kSegment, n_samples - t), d_frames, d_mix, flags, (hipStream_t)stream);
    return rc;
}

// Everything a render of up to n_samples needs except the render itself: the flattened programs on the device and the
// scratch buffers (mix partials, control tracks) at their final size.  Lets a host keep first-render set-up (a 786 MB
// hipMalloc at the headline size) out of its real-time / timed path.
int device_reserve(PatchHandle& h, uint32_t n_samples, bool want_mix, uint32_t flags)
{
    int rc = ensure_program(h, flags);
    if (rc != SRACK_OK) return rc;
    if (!h.dev) {
        rc = upload_program(h);
        if (rc != SRACK_OK) return rc;
    }
    DeviceState* d = h.dev;
    const FlatProgram& P = h.prog.voice;
    const uint32_t T = std::min(n_samples, 65536u);  // one segment
    const uint32_t eff = h.prog.effective_flags | (flags & kLaunchPolicyFlags);
    const uint32_t lanes = (fm_block_shape(P, eff, T) || fm_block_x_shape(P, eff, T)) ? (uint32_t)kBlkVoices : lanes_per_wave(P.n_voices), n_waves = (P.n_voices + lanes - 1) / lanes;
    if (want_mix && P.hdr.n_planes > 0) {
        if ((rc = grow(d->d_mixpart, d->mixpart_bytes, sizeof(float) * (size_t)P.hdr.n_planes * n_waves * T)) != SRACK_OK) return rc;
        if ((rc = grow(d->d_mixgroup, d->mixgroup_bytes, sizeof(float) * (size_t)P.hdr.n_planes * kMixSplit * T)) != SRACK_OK) return rc;
    }
    if (h.prog.n_tracks > 0 && (rc = grow(d->d_tracks, d->tracks_bytes, sizeof(float) * (size_t)h.prog.n_tracks * T)) != SRACK_OK) return rc;
    const JitKernel* special = nullptr;  // compile now what the first render would otherwise compile (frames + mix, or frames only)
    bool special_ctl = false;
    if ((rc = resolve_specialized(h, h.prog.effective_flags | (flags & kLaunchPolicyFlags), T, want_mix ? 3 : 1, &special, &special_ctl)) != SRACK_OK) return rc;
    return SRACK_OK;
}

int device_kernel_ms(PatchHandle& h, double* avg_ms, int* n_launches, int reset)
{
    h.timing_armed = reset >= 0;  // the first call arms the per-launch event pairs; reset < 0 reads what there is and disarms
    double total = 0.0;
    int n = 0;
    if (h.dev) {
        for (auto& p : h.dev->timings) {
            HIP_TRY(hipEventSynchronize(p.second));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
            total += ms;
            n++;
        }
        if (reset) {
            for (auto& p : h.dev->timings) {
                h.dev->pool.push_back(p.first);
                h.dev->pool.push_back(p.second);
            }
            h.dev->timings.clear();
        }
    }
    if (avg_ms) *avg_ms = n ? total / n : 0.0;
    if (n_launches) *n_launches = n;
    return SRACK_OK;
}

// rows of the voice program's table (ctl = false) or of the control program's one-voice table
int device_read_rows(PatchHandle& h, int ctl_stage, int first_row, int n_rows, uint32_t* host_dst)
{
    const FlatProgram& P = ctl_stage >= 0 ? h.prog.ctl[(size_t)ctl_stage] : h.prog.voice;
    const size_t V = P.n_voices;
    const uint32_t* d_table = h.dev ? (ctl_stage >= 0 ? h.dev->ctl[(size_t)ctl_stage].d_table : h.dev->voice.d_table) : nullptr;
    if (d_table) {
        const int rc_tick = tick_end(h, nullptr, true);
        if (rc_tick != SRACK_OK) return rc_tick;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(host_dst, d_table + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows, hipMemcpyDeviceToHost));
    } else {  // nothing rendered yet: the initial table
        std::memcpy(host_dst, P.table.data() + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows);
    }
    return SRACK_OK;
}

const char* device_kernel_name(const PatchHandle& h) { return h.dev ? h.dev->kernel_name : ""; }
std::string device_jit_note(const PatchHandle& h) { return h.dev ? h.dev->jit_note : std::string(); }

}  // namespace srack
