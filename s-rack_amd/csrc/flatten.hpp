// flatten.hpp — plan -> op list.  The host half of "the host flattens the graph into a
// topologically-ordered op list and calls the HIP kernels".
#pragma once
#include <string>
#include <vector>

#include "graph.hpp"
#include "program.hpp"

namespace srack {

struct VoiceOverride {  // per-voice values of one module field (srack_voices_set_field_*)
    int module, field;
    std::vector<double> values;  // n_voices entries
};

enum FusedKind : int {
    FUSED_NONE = 0,
    FUSED_VOICE_CHAIN = 1,        // OSC -> VCF -> VCA, envelope = ADSR gated by an LFO OSC (patch P1's shape), all per voice
    FUSED_VOICE_CHAIN_TRACK = 2,  // OSC -> VCF -> VCA, envelope read from a control track (P1 after uniform hoisting)
    FUSED_FM_PAIR = 3,            // OSC_M (z^-1 feedback through a Multiply) -> Multiply -> OSC_C (patch P2, B = 1)
    FUSED_CTL_GATE_ENV = 4,       // control program {OSC -> ADSR -> track}: the voice-invariant half of P1
    FUSED_VOICE_CHAIN_SEQ = 5     // [MATH(track, k)] -> OSC(cv) -> VCF([cv = track]) -> VCA(track) -> OUT (+ OUTs fed by tracks): patch P3's shape
};

struct StateLoc {  // where a module's state field lives in the voice table
    int row = -1;   // first row (-1: field is not device state)
    bool f64 = false;
    bool fixed64 = false;  // two rows holding value * 2^64 as u64 (OSC_FIXED_PHASE)
    bool flag = false;
};

struct FlatProgram {
    DevProgram hdr{};
    std::vector<DevOp> ops;
    std::vector<uint32_t> table;      // [n_rows][n_voices] initial voice table (state rows, then parameter rows)
    std::vector<uint32_t> seqtab;     // sequencer grids, 64 cells per sequencer op (DevOp::aux is the dword offset)
    std::vector<float> ring_init;     // [n_rings][B] initial ring contents, same for every voice; empty = zeros
    uint32_t fv_rows = 0;             // rows (of n_voices doubles each) of the OP_FREEVERB blocks, zero-initialised
    // Device state that is not a module field, named so that srack_patch_keep_state can find it again after a re-flatten:
    // a feedback ring (the source's port) or a reverb's block.  where: 0 = `count` rows of the voice table from row `first`,
    // 1 = global ring `first` (count = buffer_size rows of floats), 2 = rows [first, first + count) of the freeverb buffer.
    struct CarryTag {
        int module, port, where;
        int64_t first, count;
    };
    std::vector<CarryTag> carry;
    std::vector<int> op_of_module;    // module index -> op index, -1 if the module cannot reach the output
    int fused = FUSED_NONE;
    int fused_variant = 0;            // kernel-specific (which oscillator port / filter port the chain uses)
    int fm_pair_x = 0;                // the FM pair's shape with the MODULATOR exact as a whole and the carrier in its default forms (csrc/approx.cpp's answer to config 4's
                                      // loop) — 1: the ring in HBM (render_fm_pair_block_x where the delay allows, render.hip; the general path elsewhere), 2: buffer_size 1
                                      // (render_fm_pair_x)
    uint32_t n_voices = 0;
    uint32_t render_flags = 0;
    std::string description;

    StateLoc locate(const Graph& g, int module, int field) const;
};

// The flattened patch: the per-voice program and, when part of the graph is voice-invariant (no
// per-voice override anywhere upstream), a control program that evaluates that part ONCE (one
// voice) into control tracks [n_tracks][T] which the voice program reads in place (input slot >= kTrackSlot).
struct FlatPair {
    FlatProgram voice;
    // The control program, valid iff n_tracks > 0; its "planes" are the tracks.  One wave evaluating one voice is a
    // pure latency chain, so a large control program is cut into UNITS of one module each: a unit reads the tracks of
    // the modules it depends on and runs `ctl_lag` chunks behind the deepest-upstream unit, all units side by side in
    // one launch (render.hip).  A small one (or one with internal feedback) stays a single unit with lag 0.
    std::vector<FlatProgram> ctl;
    std::vector<int> ctl_lag;   // per unit: dependency depth = how many chunks it trails
    int n_tracks = 0;           // rows of the track buffer
    uint32_t effective_flags = 0;  // the render flags the programs were built for (the request, plus what flatten had to add)
    std::string approx_note;       // default mode: the first-order error bound of the forms taken (approx.cpp), or why the patch went exact
    std::vector<char> in_ctl;   // per module: evaluated by the control program
    std::vector<int> ctl_stage; // per module: its unit, -1 if not in the control program
    std::string description;
};

// Internal render flag (srack_patch_keep_state): evaluate every module of the plan, as the reference's execute() does, not only those
// the output can hear — so that a module rewired into the audible graph later has the state it would have had in the reference.
constexpr uint32_t kFlattenEvalAll = 1u << 16;

// Returns SRACK_OK or an error; on success `out` is complete.
int flatten(Graph& g, uint32_t n_voices, const std::vector<VoiceOverride>& overrides, uint32_t render_flags, FlatPair& out);

}  // namespace srack
