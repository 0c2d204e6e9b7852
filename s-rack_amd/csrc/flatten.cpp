// flatten.cpp — plan -> DevOp lists + initial voice tables.
//
// What happens here, in order:
//   1. plan (graph.cpp, = plan_execution) and find the modules that can reach the OutputModule;
//      modules that cannot are not evaluated (their buffers are unobservable offline).
//   2. dead-port elimination: an oscillator / filter port nobody reads is not computed
//      (the reference always computes all three, oscillator.rs:133-149; results on the read
//      ports are unaffected).
//  2b. default mode: which of the cheaper forms (f32 PolyBLEP, contracted ladder, f32 sine / power, fixed-point phase)
//      each module may take — approx.cpp: the forms' errors times the gain from their wires to the outputs, against
//      half the contract; an unbounded gain (a loop that amplifies, a ladder near self-oscillation, a loop through an
//      event or a pitch) behind something without an exact form of its own puts the whole patch into the exact flavour.
//   3. uniform hoisting: a module whose fields carry no per-voice override and whose inputs all
//      come from such modules produces the same samples for every voice.  That sub-graph becomes
//      the CONTROL program, evaluated once (one voice) into control tracks; the VOICE program reads
//      the tracks in place (input slot >= kTrackSlot).  (The reference has one instance of
//      everything; "N voices" is this build's axis, so sharing voice-invariant work changes no sample.)
//  3b. a control program of four or more modules is cut into one unit per module, pipelined by
//      dependency depth (render.hip runs the units side by side, one chunk apart per depth).
//   4. wires whose source runs after its sink (broken feedback edges) become a ring of
//      buffer_size samples per voice: OP_DELAY_RD before the sink, OP_DELAY_WR after the source; a
//      ring starts from the source's saved output buffer when a .srk supplied one.
//   5. constant hoisting: an oscillator without CV has a constant increment
//      delta = 440 * 2^f64(val) / f64(sample_rate); it is computed here with glibc pow — the very
//      value the reference recomputes every sample (oscillator.rs:43-48,132) — per voice.
//   6. wire slots by linear scan over the op sequence, in place where an input dies at the op that
//      reads it; voice-table rows for state and per-voice parameters; interpreter tile length from a
//      residency model (LDS granules per CU, rounds of resident waves).
//   7. pattern match for the fused kernels (P1's chain per voice / with a track, the sequencer-driven
//      chain, the FM pair with a register or an HBM ring, the gate -> envelope control program).
#include "flatten.hpp"

#include "approx.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <sstream>

namespace srack {

namespace {

constexpr int kRingLdsMax = 16;         // rings up to this many samples live in LDS rows (persisted as state)
constexpr int kTileMax = 32;            // samples per interpreter tile

struct Wire {
    int def_op = -1;
    int last_use = -1;
    int slot = -1;
};

uint32_t f32_bits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

DevOp blank_op(int kind, int module)
{
    DevOp op{};
    op.kind = kind;
    op.module = module;
    op.state_row = -1;
    op.delta_row = -1;
    op.seq_row = -1;
    for (int j = 0; j < kMaxIn; j++) op.in_slot[j] = -1;
    for (int j = 0; j < kMaxOut; j++) op.out_slot[j] = -1;
    for (int j = 0; j < kMaxPar; j++) op.par_row[j] = -1;
    return op;
}

struct Analysis {  // whole-graph facts shared by both programs
    std::vector<char> live;
    std::vector<uint32_t> port_live;
    std::vector<char> in_ctl;
    std::vector<int> stage;                          // per module: control stage (>= 0) or -1 = voice program
    std::vector<char> sine_loose;                    // per oscillator: its sine port cannot reach a pitch input (OSC_SINE_LOOSE)
    std::vector<char> saw_fixed;                     // per oscillator, default mode: its saw can reach neither a pitch input nor a threshold (OSC_FIXED_PHASE where the pitch is constant)
    std::vector<char> nonlin_loose;                  // per NonLinear module: its output cannot reach a pitch input or a threshold (NONLIN_LOOSE)
    std::vector<char> osc_exact;                     // per oscillator, default mode: evaluated as the reference spells it, all of it (OSC_EXACT on that op only: approx.cpp)
    std::vector<char> exact_src;                     // per oscillator / filter, default mode: an approximated output of it can reach a pitch input
                                                     // (OSC_EXACT_BLEP / VCF_LITERAL)
    std::vector<std::pair<int, int>> tracks;         // (module, port) exported by a control stage
    std::map<std::pair<int, int>, int> track_of;
};

struct Builder {
    Graph& g;
    uint32_t V;
    const std::vector<VoiceOverride>& ov;
    uint32_t render_flags;
    const Analysis& A;
    int stage;        // which program is being built: a control stage (>= 0) or -1 = the voice program
    bool is_ctl;
    FlatProgram& out;
    std::vector<Wire> wires;
    std::vector<std::vector<uint32_t>> rows;  // row-major voice table under construction

    const VoiceOverride* find_override(int module, int field) const
    {
        const VoiceOverride* hit = nullptr;
        for (const auto& o : ov)
            if (o.module == module && o.field == field) hit = &o;  // last one wins
        return hit;
    }

    double field(int module, int f) const { return g.modules[(size_t)module].fields[(size_t)f]; }

    int new_row()
    {
        rows.emplace_back();
        return (int)rows.size() - 1;
    }

    int state_row_f32(int module, int f)
    {
        int r = new_row();
        auto& row = rows[(size_t)r];
        row.resize(V);
        if (const VoiceOverride* o = find_override(module, f))
            for (uint32_t v = 0; v < V; v++) row[v] = f32_bits((float)o->values[v]);
        else
            std::fill(row.begin(), row.end(), f32_bits((float)field(module, f)));
        return r;
    }

    int state_row_flag(int module, int f)
    {
        int r = new_row();
        auto& row = rows[(size_t)r];
        row.resize(V);
        if (const VoiceOverride* o = find_override(module, f))
            for (uint32_t v = 0; v < V; v++) row[v] = (uint32_t)(int)o->values[v];
        else
            std::fill(row.begin(), row.end(), (uint32_t)(int)field(module, f));
        return r;
    }

    int rows_f64(const std::vector<double>* per_voice, double uniform)
    {
        int lo = new_row(), hi = new_row();
        rows[(size_t)lo].resize(V);
        rows[(size_t)hi].resize(V);
        for (uint32_t v = 0; v < V; v++) {
            double d = per_voice ? (*per_voice)[v] : uniform;
            uint64_t u;
            std::memcpy(&u, &d, 8);
            rows[(size_t)lo][v] = (uint32_t)u;
            rows[(size_t)hi][v] = (uint32_t)(u >> 32);
        }
        return lo;
    }

    void param(DevOp& op, int k, int module, int f, std::vector<std::pair<int, const VoiceOverride*>>& deferred)
    {
        op.par_val[k] = (float)field(module, f);
        op.par_row[k] = -1;
        if (const VoiceOverride* o = find_override(module, f)) {
            op.par_row[k] = -2;  // patched to a real row after all state rows are allocated
            deferred.emplace_back((int)out.ops.size() * kMaxPar + k, o);
        }
    }

    // Is this wire a sequencer's note CV, possibly offset / scaled / mixed with constants?  Such a CV holds one value
    // for thousands of samples, which lets the oscillator carry its phase terms between note changes.
    bool stepwise(int module, int port, int depth) const
    {
        if (module < 0 || depth > 8) return false;
        const Module& m = g.modules[(size_t)module];
        switch (m.type) {
        case SRACK_MOD_GRID_SEQUENCER: return port == SRACK_GRIDSEQ_OUT_CV;
        case SRACK_MOD_MATH:
        case SRACK_MOD_MONO_MIXER: {
            bool any = false;
            for (const InputRef& in : m.in)
                if (in.src >= 0) {
                    if (!stepwise(in.src, in.port, depth + 1)) return false;
                    any = true;
                }
            return any;
        }
        default: return false;
        }
    }

    int build();
    void match_fused(bool has_rings);
};

int Builder::build()
{
    const int n_mod = (int)g.modules.size();
    DevProgram& H = out.hdr;
    H.n_channels = (int)g.cfg.channels;
    H.buffer_size = (int)g.cfg.buffer_size;
    for (int c = 0; c < 8; c++) H.channel_plane[c] = -1;
    out.op_of_module.assign((size_t)n_mod, -1);
    out.n_voices = V;
    out.render_flags = render_flags;
    const int output = g.plan.output;
    const auto& pos = g.plan.position;
    auto mine = [&](int m) { return A.live[(size_t)m] && A.stage[(size_t)m] == stage; };
    std::vector<int> my_tracks;  // rows of the track buffer this program reads, in order of first use
    auto is_delayed = [&](int src, int sink) { return pos[(size_t)src] > pos[(size_t)sink]; };

    // rings: one per (src, port) in this program that has a delayed reader in this program
    std::map<std::pair<int, int>, int> ring_of;
    struct Ring { int first_row, global_id, src, port; };
    std::vector<Ring> rings;
    for (int m : g.plan.order) {
        if (!mine(m)) continue;
        for (const InputRef& in : g.modules[(size_t)m].in)
            if (in.src >= 0 && mine(in.src) && is_delayed(in.src, m) && !ring_of.count({in.src, in.port})) {
                ring_of[{in.src, in.port}] = (int)rings.size();
                rings.push_back(Ring{-1, -1, in.src, in.port});
            }
    }
    const int B = (int)g.cfg.buffer_size;
    const bool rings_in_lds = B <= kRingLdsMax;

    std::map<std::pair<int, int>, int> wire_of;  // (module, port) -> wire id; in_slot / out_slot hold WIRE ids until the scan
    std::vector<std::pair<int, const VoiceOverride*>> deferred, deferred_delta;
    std::vector<int> seq_ops;  // sequencer ops: each gets one read-only row to stage its 64 cells in
    auto new_wire = [&](int def_op) {
        wires.push_back(Wire{def_op, def_op, -1});
        return (int)wires.size() - 1;
    };
    auto use_wire = [&](int w, int op_index) { wires[(size_t)w].last_use = std::max(wires[(size_t)w].last_use, op_index); };
    int n_planes = 0;

    for (int m : g.plan.order) {
        if (!mine(m)) continue;
        const Module& mod = g.modules[(size_t)m];
        int in_wire[kMaxIn];
        for (int k = 0; k < kMaxIn; k++) in_wire[k] = -1;
        for (int k = 0; k < mod.n_in; k++) {
            const InputRef& in = mod.in[(size_t)k];
            if (in.src < 0) continue;
            if (A.stage[(size_t)in.src] != stage) {  // a control track from an (earlier) control stage — never a delayed edge,
                const int tk = A.track_of.at({in.src, in.port});  // see the uniform analysis; no wire, no slot
                size_t local = std::find(my_tracks.begin(), my_tracks.end(), tk) - my_tracks.begin();
                if (local == my_tracks.size()) my_tracks.push_back(tk);
                in_wire[k] = -2 - (int)local;
            } else if (is_delayed(in.src, m)) {
                DevOp rd = blank_op(OP_DELAY_RD, in.src);
                rd.aux = ring_of.at({in.src, in.port});
                int w = new_wire((int)out.ops.size());
                rd.out_slot[0] = w;
                out.ops.push_back(rd);
                in_wire[k] = w;
            } else {
                auto it = wire_of.find({in.src, in.port});
                if (it == wire_of.end()) {
                    set_error("flatten: internal error, source wire not defined before its reader");
                    return SRACK_ERR_INVALID;
                }
                in_wire[k] = it->second;
            }
        }
        const int oi = (int)out.ops.size();
        DevOp op = blank_op(OP_NONE, m);
        for (int j = 0; j < kMaxIn; j++) op.in_slot[j] = in_wire[j];
        for (int k = 0; k < kMaxIn; k++)
            if (in_wire[k] >= 0) use_wire(in_wire[k], oi);
        auto connected = [&](int k) { return in_wire[k] != -1; };
        const uint32_t pl = A.port_live[(size_t)m];

        switch (mod.type) {
        case SRACK_MOD_OSCILLATOR: {
            op.kind = OP_OSC;
            if (connected(0)) op.flags |= OSC_HAS_CV;
            if (connected(1)) op.flags |= OSC_HAS_SYNC;
            if (connected(0) && stepwise(mod.in[0].src, mod.in[0].port, 0)) op.flags |= OSC_CV_STEPWISE;
            if (connected(0) && wire_sweeps(g, mod.in[0].src)) op.flags |= OSC_CV_AUDIO_RATE;
            if (field(m, SRACK_OSC_ANTIALIASING) != 0.0) op.flags |= OSC_AA;
            if (pl & 1u) op.flags |= OSC_OUT_SINE;
            if (pl & 2u) op.flags |= OSC_OUT_SQUARE;
            if (pl & 4u) op.flags |= OSC_OUT_SAW;
            if ((render_flags & SRACK_RENDER_EXACT_OSC) || (!A.osc_exact.empty() && A.osc_exact[(size_t)m])) op.flags |= OSC_EXACT;
            if (A.sine_loose[(size_t)m] && !(op.flags & OSC_EXACT)) op.flags |= OSC_SINE_LOOSE;
            if (A.exact_src[(size_t)m] && !(op.flags & OSC_EXACT)) op.flags |= OSC_EXACT_BLEP;
            op.sample_rate = (double)g.cfg.sample_rate;  // `self.sample_rate as f64`, u16 in the reference
            {   // pos: f64 state, two rows (lo, hi)
                const VoiceOverride* o = find_override(m, SRACK_OSC_POS);
                op.state_row = rows_f64(o ? &o->values : nullptr, field(m, SRACK_OSC_POS));
            }
            state_row_flag(m, SRACK_OSC_SYNC_LAST);
            param(op, OSC_P_VAL, m, SRACK_OSC_VAL, deferred);
            if (!(op.flags & OSC_HAS_CV)) {
                // get_freq_in_hz(None, i) / sample_rate — loop-invariant, same bits as the reference
                op.delta = 440.0 * std::pow(2.0, (double)(float)field(m, SRACK_OSC_VAL)) / op.sample_rate;
                bool small = op.delta < 0.25;
                if (const VoiceOverride* o = find_override(m, SRACK_OSC_VAL)) {
                    op.delta_row = -2;
                    deferred_delta.emplace_back(oi, o);
                    for (uint32_t v = 0; v < V && small; v++)
                        small = 440.0 * std::pow(2.0, (double)(float)o->values[v]) / op.sample_rate < 0.25;
                }
                const uint32_t ports = op.flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
                if (small && !(op.flags & (OSC_HAS_SYNC | OSC_EXACT_BLEP)) && (op.flags & OSC_AA) && ports && !(ports & (ports - 1)))
                    op.flags |= (op.flags & OSC_EXACT) ? OSC_CONST_SMALL : (OSC_CONST_SMALL | OSC_CONST_FAST);
            }
            break;
        }
        case SRACK_MOD_MOOG_FILTER:
            op.kind = OP_VCF;
            if (connected(0)) op.flags |= VCF_HAS_AUDIO;
            if (connected(1)) op.flags |= VCF_HAS_CV;
            if (pl & 1u) op.flags |= VCF_OUT_LP;
            if (pl & 2u) op.flags |= VCF_OUT_BP;
            if (pl & 4u) op.flags |= VCF_OUT_HP;
            if (A.exact_src[(size_t)m] && !(render_flags & SRACK_RENDER_EXACT_OSC)) op.flags |= VCF_LITERAL;
            op.state_row = state_row_f32(m, SRACK_VCF_ST_F);
            state_row_f32(m, SRACK_VCF_ST_P);
            state_row_f32(m, SRACK_VCF_ST_Q);
            for (int k = 0; k < 5; k++) state_row_f32(m, SRACK_VCF_ST_B0 + k);
            state_row_f32(m, SRACK_VCF_ST_FREQ);
            state_row_f32(m, SRACK_VCF_ST_RES);
            param(op, VCF_P_FREQ, m, SRACK_VCF_FREQ, deferred);
            param(op, VCF_P_RES, m, SRACK_VCF_RES, deferred);
            param(op, VCF_P_EXP, m, SRACK_VCF_EXP_AMT, deferred);
            break;
        case SRACK_MOD_ADSR:
            op.kind = OP_ADSR;
            if (connected(0)) op.flags |= ADSR_HAS_GATE;
            op.state_row = state_row_f32(m, SRACK_ADSR_PHASE);
            state_row_flag(m, SRACK_ADSR_MODE);
            state_row_f32(m, SRACK_ADSR_R_VAL);
            state_row_f32(m, SRACK_ADSR_FROM_A_VAL);
            state_row_flag(m, SRACK_ADSR_GATE_LAST);
            param(op, ADSR_P_A, m, SRACK_ADSR_A_SEC, deferred);
            param(op, ADSR_P_D, m, SRACK_ADSR_D_SEC, deferred);
            param(op, ADSR_P_S, m, SRACK_ADSR_S_VAL, deferred);
            param(op, ADSR_P_R, m, SRACK_ADSR_R_SEC, deferred);
            param(op, ADSR_P_SR, m, SRACK_ADSR_SAMPLE_RATE, deferred);
            break;
        case SRACK_MOD_VCA:
            op.kind = OP_VCA;
            if (connected(0)) op.flags |= VCA_HAS_AUDIO;
            if (connected(1)) op.flags |= VCA_HAS_CV;
            param(op, VCA_P_NEG, m, SRACK_VCA_NEGATIVE, deferred);
            break;
        case SRACK_MOD_MONO_MIXER:
            op.kind = OP_MIX;
            for (int k = 0; k < 4; k++) {
                if (connected(k)) op.flags |= 1u << k;
                param(op, MIX_P_GAIN0 + k, m, SRACK_MIX_GAIN0 + k, deferred);
            }
            break;
        case SRACK_MOD_MATH:
            op.kind = OP_MATH;
            if (connected(0)) op.flags |= MATH_HAS_IN1;
            if (connected(1)) op.flags |= MATH_HAS_IN2;
            op.flags |= ((uint32_t)(int)field(m, SRACK_MATH_OPERATION) & 3u) << MATH_OP_SHIFT;
            param(op, MATH_P_CONST, m, SRACK_MATH_CONSTANT, deferred);
            break;
        case SRACK_MOD_NONLINEAR:
            op.kind = OP_NONLIN;
            if (connected(0)) op.flags |= MATH_HAS_IN1;
            if (connected(1)) op.flags |= MATH_HAS_IN2;
            if (render_flags & SRACK_RENDER_EXACT_OSC) op.flags |= NONLIN_EXACT;
            else if (A.nonlin_loose[(size_t)m]) op.flags |= NONLIN_LOOSE;
            param(op, NONLIN_P_CONST, m, SRACK_NONLIN_CONSTANT, deferred);
            break;
        case SRACK_MOD_FREEVERB: {  // freeverb.rs + the freeverb crate's Freeverb::new / set_* (restated; see oracle/srack_oracle.c)
            op.kind = OP_FREEVERB;
            for (int f = 0; f < SRACK_FREEVERB__NFIELDS; f++)
                if (find_override(m, f)) {
                    set_error("flatten: per-voice overrides of FreeverbModule parameters are not supported");
                    return SRACK_ERR_UNSUPPORTED;
                }
            static const uint32_t comb_tuning[8] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617}, allpass_tuning[4] = {556, 441, 341, 225};
            const uint32_t sr = g.cfg.sample_rate;
            uint32_t len[kFvLines], first[kFvLines], total = 0;
            for (int j = 0; j < kFvLines; j++) {
                const uint32_t tuning = (j < 16 ? comb_tuning[j / 2] : allpass_tuning[(j - 16) / 2]) + ((j & 1) ? 23u : 0u);
                len[j] = (uint32_t)((double)tuning * (double)sr / 44100.0);  // adjust_length: `(length as f64 * sr as f64 / 44100.0) as usize`
                if (len[j] < 4) {  // (a zero-length line panics in the crate; the tile function reads four samples of a line at a time)
                    set_error("flatten: FreeverbModule needs a sample rate of at least 784 Hz");
                    return SRACK_ERR_UNSUPPORTED;
                }
                first[j] = total;
                total += len[j];
            }
            // Freeverb::new's defaults, then set_freeverb(all = true) in its order (freeverb.rs:88-114): every setter runs once
            const double dampening = field(m, SRACK_FREEVERB_DAMPENING) * 0.4, room = field(m, SRACK_FREEVERB_ROOM_SIZE) * 0.28 + 0.7;
            const bool frozen = field(m, SRACK_FREEVERB_FREEZE) != 0.0;
            const double wet = field(m, SRACK_FREEVERB_WET) * 3.0, width = field(m, SRACK_FREEVERB_WIDTH);
            const double comb_damp = frozen ? 0.0 : dampening;
            const double par[7] = {frozen ? 1.0 : room, comb_damp, 1.0 - comb_damp, wet * (width / 2.0 + 0.5), wet * ((1.0 - width) / 2.0),
                                   field(m, SRACK_FREEVERB_DRY), 1.0 /* input_gain: set by new(); the public set_freeze leaves it alone */};
            op.aux = (int)out.seqtab.size();
            out.seqtab.insert(out.seqtab.end(), len, len + kFvLines);
            out.seqtab.insert(out.seqtab.end(), first, first + kFvLines);
            for (double d : par) {
                uint64_t u;
                std::memcpy(&u, &d, 8);
                out.seqtab.push_back((uint32_t)u);
                out.seqtab.push_back((uint32_t)(u >> 32));
            }
            op.delta_row = (int)out.fv_rows;
            out.carry.push_back(FlatProgram::CarryTag{m, 0, 2, (int64_t)out.fv_rows, (int64_t)kFvStates + total});
            out.fv_rows += (uint32_t)kFvStates + total;
            break;
        }
        case SRACK_MOD_NOISE: {  // stateless: sample n of voice v is a function of (seed, module, first_voice + v, n)
            op.kind = OP_NOISE;
            const uint64_t base = noise_base_key(g.cfg.noise_seed, m), first = g.cfg.noise_first_voice;
            std::memcpy(&op.delta, &base, 8);
            std::memcpy(&op.sample_rate, &first, 8);
            break;
        }
        case SRACK_MOD_SAMPLE: {
            op.kind = OP_SAMPLE;
            if (connected(0)) op.flags |= SMP_HAS_GATE;
            if (connected(1)) op.flags |= SMP_HAS_CV;
            op.state_row = state_row_f32(m, SRACK_SAMPLE_POS);
            const int playing_row = state_row_flag(m, SRACK_SAMPLE_PLAYING);
            state_row_flag(m, SRACK_SAMPLE_GATE_LAST);
            if (field(m, SRACK_SAMPLE_WAVE_NEW) != 0.0) {  // `if wavebox.new { pos = 0.0; playing = false; }` at the first calc (sample.rs:206-210)
                rows[(size_t)op.state_row].assign(V, 0u);
                rows[(size_t)playing_row].assign(V, 0u);
            }
            param(op, SMP_P_SR, m, SRACK_SAMPLE_SAMPLE_RATE, deferred);
            param(op, SMP_P_WAVE_SR, m, SRACK_SAMPLE_WAVE_SAMPLE_RATE, deferred);
            if (mod.wave.size() >= (size_t)1 << 31) {
                set_error("flatten: wave longer than 2^31 - 1 samples");
                return SRACK_ERR_UNSUPPORTED;
            }
            op.seq_len = (int)mod.wave.size();
            op.aux = (int)out.seqtab.size();
            for (float f : mod.wave) out.seqtab.push_back(f32_bits(f));
            break;
        }
        case SRACK_MOD_GRID_SEQUENCER:
        case SRACK_MOD_PATTERN_SEQUENCER: {
            const bool grid = mod.type == SRACK_MOD_GRID_SEQUENCER;
            op.kind = grid ? OP_GRIDSEQ : OP_PATSEQ;
            op.flags = pl & 0x1ffu;  // which output ports are read
            if (connected(0)) op.flags |= SEQ_HAS_STEP;
            if (connected(1)) op.flags |= SEQ_HAS_SYNC;
            op.state_row = state_row_flag(m, grid ? SRACK_GRIDSEQ_CURRENT_STEP : SRACK_PATSEQ_CURRENT_STEP);
            state_row_flag(m, grid ? SRACK_GRIDSEQ_STEP_LAST : SRACK_PATSEQ_STEP_LAST);
            state_row_flag(m, grid ? SRACK_GRIDSEQ_SYNC_LAST : SRACK_PATSEQ_SYNC_LAST);
            if (grid) {
                state_row_f32(m, SRACK_GRIDSEQ_LAST);
                param(op, GRIDSEQ_P_SPO, m, SRACK_GRIDSEQ_STEPS_PER_OCTAVE, deferred);
            }
            op.seq_len = (int)field(m, grid ? SRACK_GRIDSEQ_LENGTH : SRACK_PATSEQ_LENGTH);
            op.aux = (int)out.seqtab.size();
            out.seqtab.insert(out.seqtab.end(), mod.cells.begin(), mod.cells.end());
            seq_ops.push_back(oi);
            break;
        }
        case SRACK_MOD_OUTPUT:
            op.kind = OP_OUT;
            break;
        default:
            set_error("flatten: unsupported module type");
            return SRACK_ERR_UNSUPPORTED;
        }

        if (mod.type == SRACK_MOD_OUTPUT) {
            if (m != output) continue;  // only the plan's sink is observable
            // one OP_OUT per distinct source wire (plane); channels map onto planes
            std::map<int, int> plane_of_wire;
            for (int c = 0; c < mod.n_in && c < 8; c++) {
                if (in_wire[c] == -1) continue;
                auto it = plane_of_wire.find(in_wire[c]);
                if (it == plane_of_wire.end()) {
                    DevOp o2 = blank_op(OP_OUT, m);
                    o2.in_slot[0] = in_wire[c];
                    o2.aux = n_planes;
                    if (in_wire[c] >= 0) use_wire(in_wire[c], (int)out.ops.size());
                    out.ops.push_back(o2);
                    it = plane_of_wire.emplace(in_wire[c], n_planes++).first;
                }
                H.channel_plane[c] = it->second;
            }
            out.op_of_module[(size_t)m] = oi;
            continue;
        }

        for (int p = 0; p < mod.n_out; p++)
            if (pl & (1u << p)) {
                int w = new_wire(oi);
                op.out_slot[p] = w;
                wire_of[{m, p}] = w;
            }
        out.op_of_module[(size_t)m] = oi;
        out.ops.push_back(op);
        for (int p = 0; p < mod.n_out; p++) {
            auto it = ring_of.find({m, p});  // source side of a ring fed by this module
            if (it != ring_of.end()) {
                DevOp wr = blank_op(OP_DELAY_WR, m);
                wr.aux = it->second;
                wr.in_slot[0] = wire_of.at({m, p});
                use_wire(wr.in_slot[0], (int)out.ops.size());
                out.ops.push_back(wr);
            }
            if (is_ctl) {  // control program: exported wires are written out as track planes
                auto tk = A.track_of.find({m, p});
                if (tk != A.track_of.end()) {
                    DevOp o2 = blank_op(OP_OUT, m);
                    o2.in_slot[0] = wire_of.at({m, p});
                    o2.aux = tk->second;
                    use_wire(o2.in_slot[0], (int)out.ops.size());
                    out.ops.push_back(o2);
                    n_planes = std::max(n_planes, tk->second + 1);
                }
            }
        }
    }
    if ((int)out.ops.size() > kMaxOps) {
        set_error("flatten: patch needs more than " + std::to_string(kMaxOps) + " ops");
        return SRACK_ERR_UNSUPPORTED;
    }

    // ---- rings: storage ---------------------------------------------------------------------
    // Initial contents: the source's output buffer as it is before the first tick — zeros for a fresh module
    // (synth.rs:31-33), the saved block for a patch loaded from a .srk (Module::out_init).
    int n_global = 0;
    for (Ring& r : rings) {
        const Module& src = g.modules[(size_t)r.src];
        const std::vector<float>* init = (size_t)r.port < src.out_init.size() && src.out_init[(size_t)r.port].size() == (size_t)B ? &src.out_init[(size_t)r.port] : nullptr;
        if (rings_in_lds) {
            r.first_row = (int)rows.size();
            for (int k = 0; k < B; k++) rows[(size_t)new_row()].assign(V, init ? f32_bits((*init)[(size_t)k]) : 0u);
        } else {
            r.global_id = n_global++;
            if (init) {
                out.ring_init.resize((size_t)n_global * B, 0.0f);
                std::copy(init->begin(), init->end(), out.ring_init.begin() + (size_t)r.global_id * B);
            }
        }
    }
    if (!out.ring_init.empty()) out.ring_init.resize((size_t)n_global * B, 0.0f);
    for (const Ring& r : rings) out.carry.push_back(FlatProgram::CarryTag{r.src, r.port, rings_in_lds ? 0 : 1, rings_in_lds ? r.first_row : r.global_id, B});
    for (DevOp& op : out.ops)
        if (op.kind == OP_DELAY_RD || op.kind == OP_DELAY_WR) {
            const Ring& r = rings[(size_t)op.aux];
            if (rings_in_lds) {
                op.aux = r.first_row;
            } else {
                op.aux = r.global_id;
                op.flags |= DELAY_RING_GLOBAL;
            }
        }
    H.n_rings = n_global;
    H.n_state_rows = (int)rows.size();

    // ---- per-voice parameter rows (read-only) -------------------------------------------------
    for (auto& d : deferred) {
        DevOp& op = out.ops[(size_t)(d.first / kMaxPar)];
        int r = new_row();
        auto& row = rows[(size_t)r];
        row.resize(V);
        for (uint32_t v = 0; v < V; v++) row[v] = f32_bits((float)d.second->values[v]);
        op.par_row[d.first % kMaxPar] = r;
    }
    for (auto& d : deferred_delta) {
        DevOp& op = out.ops[(size_t)d.first];
        std::vector<double> delta(V);
        for (uint32_t v = 0; v < V; v++) delta[v] = 440.0 * std::pow(2.0, (double)(float)d.second->values[v]) / op.sample_rate;
        op.delta_row = rows_f64(&delta, 0.0);
    }
    for (int oi : seq_ops) {  // contents are written by the tile function; the row only reserves LDS
        int r = new_row();
        rows[(size_t)r].assign(V, 0u);
        out.ops[(size_t)oi].seq_row = r;
    }
    H.n_rows = (int)rows.size();

    // ---- wire slots: linear scan ----------------------------------------------------------------
    {
        std::vector<int> free_slots;
        int n_slots = 0;
        std::vector<std::vector<int>> expire(out.ops.size() + 1), defs(out.ops.size() + 1);
        for (size_t w = 0; w < wires.size(); w++) {
            expire[(size_t)wires[w].last_use].push_back((int)w);
            defs[(size_t)wires[w].def_op].push_back((int)w);
        }
        for (size_t i = 0; i < out.ops.size(); i++) {
            // An input read for the last time by this op hands its slot to one of the op's outputs: every tile function
            // reads a group of samples from all its inputs before it writes that group's outputs (tile_run), and sample
            // i of a wire is only ever touched at index i, so computing in place is safe — and each slot saved is a
            // tile of LDS, i.e. longer tiles or more resident waves.
            for (int w : expire[i])
                if (wires[(size_t)w].def_op < (int)i) free_slots.push_back(wires[(size_t)w].slot);
            for (int w : defs[i]) {
                if (free_slots.empty()) {
                    wires[(size_t)w].slot = n_slots++;
                } else {
                    wires[(size_t)w].slot = free_slots.back();
                    free_slots.pop_back();
                }
            }
            for (int w : expire[i])
                if (wires[(size_t)w].def_op == (int)i) free_slots.push_back(wires[(size_t)w].slot);
        }
        H.n_slots = n_slots;
        for (DevOp& op : out.ops) {
            for (int k = 0; k < kMaxIn; k++) {
                if (op.in_slot[k] >= 0)
                    op.in_slot[k] = wires[(size_t)op.in_slot[k]].slot;
                else if (op.in_slot[k] <= -2)
                    op.in_slot[k] = kTrackSlot + (-2 - op.in_slot[k]);
            }
            for (int k = 0; k < kMaxOut; k++)
                if (op.out_slot[k] >= 0) op.out_slot[k] = wires[(size_t)op.out_slot[k]].slot;
        }
    }
    H.n_ops = (int)out.ops.size();
    H.n_planes = n_planes;
    if ((int)my_tracks.size() > kMaxTracksRead) {
        set_error("flatten: one program reads more than " + std::to_string(kMaxTracksRead) + " control tracks");
        return SRACK_ERR_UNSUPPORTED;
    }
    H.n_tracks = (int)my_tracks.size();
    for (size_t k = 0; k < my_tracks.size(); k++) H.track_id[k] = my_tracks[k];

    // ---- tile length: LDS per wave decides how many waves a CU holds ---------------------------------------
    // One wave needs (fixed rows + slots x tile) x 256 B of LDS, handed out in 1280-B granules from 160 KB per CU.
    // A wave's module chains are latency-bound, so a round of resident waves costs nearly the same whether the CU
    // holds 4 or 15 of them; what costs is another round, and a tile's fixed work (op fields, state rows in and out,
    // the call) paid more often when tiles are short.  The model is fitted to P1 through the interpreter on MI355X
    // (measured with SRACK_TILE_MAX sweeps in round 1): a round with r waves per CU takes 17 + 0.7 r (ms at 48 000 samples), the last round holds
    // the remainder, a tile's fixed work is worth ~8 samples, and a CU that would be exactly full spills a few waves
    // into an extra, nearly empty round.  It predicts the measured times of eleven (tile, voices) pairs within 15 %
    // after a common scale, which is all a choice between tile lengths needs.
    {
        const int fixed_rows = H.n_rows + 2 + H.n_tracks;  // voice table + zero and trash rows + one row per control track
        auto lds_bytes = [&](int tile) { return (fixed_rows + H.n_slots * tile) * 256; };
        int tile_max = kTileMax;
        if (const char* e = getenv("SRACK_TILE_MAX")) tile_max = std::min(std::max(atoi(e), 1), 64);  // tuning knob (tools/), clamped
        if (!rings.empty()) tile_max = std::min(tile_max, B);              // a tile may not span more than one ring period
        int tile = 0;
        if (is_ctl) {  // one wave: residency is irrelevant, long tiles are not (a row holds 64 samples of a track: the cap)
            if (!getenv("SRACK_TILE_MAX")) tile_max = rings.empty() ? 64 : std::min(64, B);
            int budget = 56 * 1024;
            if (const char* e = getenv("SRACK_LDS_BUDGET")) budget = std::min(std::max(atoi(e), 1024), 64 * 1024);
            for (tile = std::max(tile_max, 1); tile > 1 && lds_bytes(tile) > budget; tile >>= 1) {}
        } else {
            constexpr int kCUs = 256, kLdsPerCU = 160 * 1024, kGranule = 1280, kWaveSlots = 20;  // 5 waves per SIMD (<= 96 VGPRs)
            const int need = (int)(((V + 63) / 64 + kCUs - 1) / kCUs);  // waves per CU this render asks for
            double best = 0.0;
            static const int kCandidates[] = {32, 28, 24, 20, 16, 12, 8, 4, 2, 1};
            for (int t : kCandidates) {
                if (t > tile_max && t != 1) continue;
                const int bytes = lds_bytes(t);
                if (bytes > 64 * 1024) continue;
                const int alloc = (bytes + kGranule - 1) / kGranule * kGranule;
                const int cap = std::min(kWaveSlots, kLdsPerCU / alloc);
                const int full = need / cap, rem = need % cap;
                double rounds_ms = full * (17.0 + 0.7 * cap) + (rem ? 17.0 + 0.7 * rem : 0.0);
                if (need == cap) rounds_ms += 8.0;
                const double cost = rounds_ms * (1.0 + 8.0 / t);
                if (tile == 0 || cost < best) {
                    best = cost;
                    tile = t;
                }
            }
        }
        if (tile < 1 || lds_bytes(tile) > 64 * 1024) {
            set_error("flatten: patch state does not fit the LDS budget of the tile interpreter");
            return SRACK_ERR_UNSUPPORTED;
        }
        H.tile = tile;
    }

    out.table.resize((size_t)H.n_rows * V);
    for (int r = 0; r < H.n_rows; r++) std::memcpy(&out.table[(size_t)r * V], rows[(size_t)r].data(), sizeof(uint32_t) * V);

    // Default mode, a voice program's constant-pitch saw that nothing integrates or thresholds (A.saw_fixed): every kernel keeps its phase in
    // 64-bit fixed point (modules.hip.h, FOsc — three integer-rate instructions per sample where the f64 accumulator takes three at half
    // rate or less).  State and increment change representation here, once: phase * 2^64 as u64 in the same two rows.
    if (!is_ctl && !A.saw_fixed.empty())
        for (DevOp& osc : out.ops) {
            if (osc.kind != OP_OSC || !(osc.flags & OSC_CONST_FAST) || (osc.flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW)) != OSC_OUT_SAW || !A.saw_fixed[(size_t)osc.module]) continue;
            auto to_fixed = [](double x) {  // x in [0, 1): exact whenever x has no bits below 2^-64
                const double y = std::ldexp(x - std::floor(x), 64);
                return y >= 18446744073709551616.0 ? ~0ull : (uint64_t)y;
            };
            auto convert_rows = [&](int lo_row) {
                uint32_t* lo = &out.table[(size_t)lo_row * V];
                uint32_t* hi = &out.table[(size_t)(lo_row + 1) * V];
                for (uint32_t v = 0; v < V; v++) {
                    const uint64_t bits = ((uint64_t)hi[v] << 32) | lo[v];
                    double d;
                    std::memcpy(&d, &bits, 8);
                    const uint64_t f = to_fixed(d);
                    lo[v] = (uint32_t)f;
                    hi[v] = (uint32_t)(f >> 32);
                }
            };
            osc.flags |= OSC_FIXED_PHASE;
            convert_rows(osc.state_row + OSC_S_POS_LO);
            if (osc.delta_row >= 0) {
                convert_rows(osc.delta_row);
            } else {
                const uint64_t f = to_fixed(osc.delta);
                std::memcpy(&osc.delta, &f, 8);
            }
        }
    if (!(render_flags & SRACK_RENDER_NO_FUSION)) match_fused(!rings.empty());
    if (is_ctl && !(render_flags & SRACK_RENDER_NO_FUSION) && rings.empty() && H.n_ops == 3 && out.ops[0].kind == OP_OSC &&
        out.ops[1].kind == OP_ADSR && out.ops[2].kind == OP_OUT && (out.ops[0].flags & OSC_CONST_SMALL) && (out.ops[1].flags & ADSR_HAS_GATE) &&
        (!(out.ops[0].flags & OSC_EXACT) || (render_flags & SRACK_RENDER_EXACT_OSC)) &&
        g.modules[(size_t)out.ops[1].module].in[0].src == out.ops[0].module && out.ops[2].module == out.ops[1].module)
        out.fused = FUSED_CTL_GATE_ENV;  // uniform parameters only (V == 1, no overrides): par_val / delta are used directly

    std::ostringstream d;
    d << (is_ctl ? "ctl[" : "voice[") << "ops=" << H.n_ops << " slots=" << H.n_slots << " rows=" << H.n_rows << " (state " << H.n_state_rows
      << ") planes=" << H.n_planes << " rings=" << rings.size() << (rings.empty() ? "" : (rings_in_lds ? "(lds)" : "(hbm)")) << " tile=" << H.tile
      << " fused=" << out.fused << "]";
    out.description = d.str();
    return SRACK_OK;
}

// Fused kernels (render.hip) for patch P1's shape.  All-per-voice form:
//   {OSC_A, OSC_L, VCF, ADSR, VCA, OUT}: VCF <- OSC_A, ADSR <- OSC_L, VCA <- (VCF, ADSR), one plane <- VCA.
// After uniform hoisting the LFO/ADSR pair lives in the control program and the voice program is
//   {OSC_A, VCF, VCA, OUT}: VCA <- (VCF, control track).
// Oscillators must be OSC_CONST_FAST (or the exact flavour of the same shape), filters have no CV.
void Builder::match_fused(bool has_rings)
{
    // (the fused kernels take the exact flavour as ONE template parameter: a program in which single oscillators are exact — approx.cpp's
    // answer to an unbounded gain — is the general path's, but for the one mixed shape that has a kernel of its own: fm_pair_x below)
    bool mixed = false;
    if (!(render_flags & SRACK_RENDER_EXACT_OSC))
        for (const DevOp& op : out.ops)
            if (op.kind == OP_OSC && (op.flags & OSC_EXACT)) mixed = true;
    if (has_rings) {
        // 2-operator FM with a one-sample feedback edge (patch P2 at buffer_size 1):
        //   DELAY_RD -> MATH_FB -> OSC_M -> DELAY_WR ; OSC_M -> MATH_IDX -> OSC_C -> OUT ; only sine ports, no sync
        const auto& o = out.ops;
        const DevProgram& H = out.hdr;
        auto is_scale = [](const DevOp& op) {  // Multiply by the module's constant: in1 connected, in2 not
            return op.kind == OP_MATH && (op.flags & (MATH_HAS_IN1 | MATH_HAS_IN2)) == MATH_HAS_IN1 && ((op.flags >> MATH_OP_SHIFT) & 3u) == SRACK_MATH_MULTIPLY;
        };
        auto is_fm_osc = [](const DevOp& op) {
            return op.kind == OP_OSC && (op.flags & (OSC_HAS_CV | OSC_HAS_SYNC | OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW)) == (OSC_HAS_CV | OSC_OUT_SINE);
        };
        // ... at buffer_size 1 (the delayed sine is last tick's: a register) or >= 32 (a ring in HBM whose reads a 32-sample
        // tile can issue up front); in between, a tile would read what it has just written
        const bool z1 = H.buffer_size == 1 && !(o[0].flags & DELAY_RING_GLOBAL), far = H.buffer_size >= 32 && (o[0].flags & DELAY_RING_GLOBAL);
        if ((z1 || far) && H.n_ops == 7 && H.n_planes == 1 && o[0].kind == OP_DELAY_RD && is_scale(o[1]) &&
            is_fm_osc(o[2]) && o[3].kind == OP_DELAY_WR && is_scale(o[4]) && is_fm_osc(o[5]) && o[6].kind == OP_OUT && o[0].module == o[2].module &&
            o[1].in_slot[0] == o[0].out_slot[0] && o[2].in_slot[0] == o[1].out_slot[0] && o[3].in_slot[0] == o[2].out_slot[0] &&
            o[4].in_slot[0] == o[2].out_slot[0] && o[5].in_slot[0] == o[4].out_slot[0] && o[6].in_slot[0] == o[5].out_slot[0] && o[0].aux == o[3].aux)
        {
            if (!mixed) {
                out.fused = FUSED_FM_PAIR;
                out.fused_variant = far ? 1 : 0;  // 1: ring in HBM
            } else if ((o[2].flags & OSC_EXACT) && !(o[5].flags & (OSC_EXACT | OSC_EXACT_BLEP)) && (o[5].flags & OSC_SINE_LOOSE)) {
                out.fm_pair_x = far ? 1 : 2;  // the modulator exact as a whole, the carrier — its sine only heard — in its default forms (`fused` stays FUSED_NONE)
            }
        }
        return;
    }
    if (mixed) return;
    if (is_ctl) return;  // the voice-chain shapes below are per-voice programs
    int n_kind[kOpKinds] = {0};
    for (const DevOp& op : out.ops) n_kind[op.kind]++;
    const DevProgram& H = out.hdr;
    // Sequencer-driven subtractive voice (patch P3 after hoisting): the note CV, the filter envelope and the amplitude
    // envelope arrive as control tracks; per voice there is an optional transpose (Math on the note track), the
    // oscillator, the filter and the VCA.  Further output channels may carry tracks as they are (a raw gate, ...).
    if (n_kind[OP_OSC] == 1 && n_kind[OP_VCF] == 1 && n_kind[OP_VCA] == 1 && n_kind[OP_OUT] >= 1 && n_kind[OP_OUT] <= 5 && n_kind[OP_MATH] <= 1 &&
        H.n_ops == 3 + n_kind[OP_OUT] + n_kind[OP_MATH] && !(render_flags & SRACK_RENDER_EXACT_OSC)) {
        const DevOp *math = nullptr, *osc = nullptr, *vcf = nullptr, *vca = nullptr;
        for (const DevOp& op : out.ops) {
            if (op.kind == OP_MATH) math = &op;
            if (op.kind == OP_OSC) osc = &op;
            if (op.kind == OP_VCF) vcf = &op;
            if (op.kind == OP_VCA) vca = &op;
        }
        auto src_of = [&](int sink_module, int k) { return g.modules[(size_t)sink_module].in[(size_t)k]; };
        auto is_track = [](int slot) { return slot >= kTrackSlot; };
        const uint32_t ports = osc->flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
        const uint32_t vcf_ports = vcf->flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP);
        bool ok = (osc->flags & ~(OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW | OSC_SINE_LOOSE)) == (OSC_HAS_CV | OSC_CV_STEPWISE | OSC_AA)  /* (no OSC_EXACT_BLEP) */ && ports && !(ports & (ports - 1));
        if (ok && math)
            ok = src_of(osc->module, 0).src == math->module && (math->flags & (MATH_HAS_IN1 | MATH_HAS_IN2)) == MATH_HAS_IN1 && is_track(math->in_slot[0]);
        else if (ok)
            ok = is_track(osc->in_slot[0]);
        ok = ok && !(vcf->flags & VCF_LITERAL) && (vcf->flags & VCF_HAS_AUDIO) && src_of(vcf->module, 0).src == osc->module && vcf_ports && !(vcf_ports & (vcf_ports - 1)) &&
             (!(vcf->flags & VCF_HAS_CV) || is_track(vcf->in_slot[1]));
        ok = ok && vca->flags == (VCA_HAS_AUDIO | VCA_HAS_CV) && src_of(vca->module, 0).src == vcf->module && is_track(vca->in_slot[1]);
        int n_main = 0;
        for (const DevOp& op : out.ops) {
            if (op.kind != OP_OUT || is_track(op.in_slot[0])) continue;
            n_main++;  // a plane fed by a wire: it must be the VCA's
            bool from_vca = false;
            for (int c = 0; c < H.n_channels; c++)
                if (H.channel_plane[c] == op.aux) from_vca = src_of(op.module, c).src == vca->module;
            ok = ok && from_vca;
        }
        if (ok && n_main == 1) {
            out.fused = FUSED_VOICE_CHAIN_SEQ;
            return;
        }
    }
    const bool full = H.n_ops == 6 && n_kind[OP_OSC] == 2 && n_kind[OP_VCF] == 1 && n_kind[OP_ADSR] == 1 && n_kind[OP_VCA] == 1 && n_kind[OP_OUT] == 1;
    const bool tracked = H.n_ops == 4 && n_kind[OP_OSC] == 1 && n_kind[OP_VCF] == 1 && n_kind[OP_VCA] == 1 && n_kind[OP_OUT] == 1;
    if (!full && !tracked) return;
    const DevOp *vcf = nullptr, *adsr = nullptr, *vca = nullptr, *outp = nullptr;
    for (const DevOp& op : out.ops) {
        if (op.kind == OP_VCF) vcf = &op;
        if (op.kind == OP_ADSR) adsr = &op;
        if (op.kind == OP_VCA) vca = &op;
        if (op.kind == OP_OUT) outp = &op;
    }
    auto src_of = [&](int sink_module, int k) { return g.modules[(size_t)sink_module].in[(size_t)k]; };
    auto osc_ok = [&](int module) {
        if (g.modules[(size_t)module].type != SRACK_MOD_OSCILLATOR || out.op_of_module[(size_t)module] < 0) return false;
        const DevOp& o = out.ops[(size_t)out.op_of_module[(size_t)module]];
        if (o.flags & (OSC_HAS_CV | OSC_HAS_SYNC)) return false;
        const uint32_t ports = o.flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
        if (!ports || (ports & (ports - 1)) || !(o.flags & OSC_AA)) return false;
        return (o.flags & OSC_CONST_FAST) || (o.flags & OSC_EXACT);
    };
    const uint32_t vcf_ports = vcf->flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP);
    bool ok = vca->flags == (VCA_HAS_AUDIO | VCA_HAS_CV) && src_of(vca->module, 0).src == vcf->module && src_of(outp->module, 0).src >= 0 &&
              (vcf->flags & (VCF_HAS_AUDIO | VCF_HAS_CV | VCF_LITERAL)) == VCF_HAS_AUDIO && vcf_ports && !(vcf_ports & (vcf_ports - 1)) &&
              osc_ok(src_of(vcf->module, 0).src);
    for (int c = 0; c < H.n_channels && ok; c++) {
        const InputRef& in = src_of(outp->module, c);
        ok = in.src < 0 || in.src == vca->module;
    }
    if (!ok) return;
    if (full) {
        ok = src_of(vca->module, 1).src == adsr->module && (adsr->flags & ADSR_HAS_GATE) && osc_ok(src_of(adsr->module, 0).src) &&
             src_of(adsr->module, 0).src != src_of(vcf->module, 0).src && src_of(adsr->module, 0).port == SRACK_OSC_OUT_SQUARE;
        if (ok) out.fused = FUSED_VOICE_CHAIN;
    } else {
        ok = A.in_ctl[(size_t)src_of(vca->module, 1).src] && vca->in_slot[1] >= kTrackSlot;
        // render_voice_chain_track decides at COMPILE time that a default-mode saw keeps its phase in 64-bit fixed point: the shape is only
        // this kernel's when the rows were converted above (OSC_FIXED_PHASE) — a saw that something integrates or thresholds, whose phase
        // stays an f64, renders through the general path
        const DevOp& osc = out.ops[(size_t)out.op_of_module[(size_t)src_of(vcf->module, 0).src]];
        if (!(osc.flags & OSC_EXACT) && (osc.flags & OSC_OUT_SAW) && !(osc.flags & OSC_FIXED_PHASE)) ok = false;
        if (ok) out.fused = FUSED_VOICE_CHAIN_TRACK;
    }
}

}  // namespace

StateLoc FlatProgram::locate(const Graph& g, int module, int field) const
{
    StateLoc loc;
    if (module < 0 || module >= (int)op_of_module.size()) return loc;
    int oi = op_of_module[(size_t)module];
    if (oi < 0) return loc;
    const DevOp& op = ops[(size_t)oi];
    int type = g.modules[(size_t)module].type;
    if (!Graph::field_is_state(type, field) || op.state_row < 0) return loc;
    switch (type) {
    case SRACK_MOD_OSCILLATOR:
        if (field == SRACK_OSC_POS) {
            loc.row = op.state_row + OSC_S_POS_LO;
            loc.f64 = true;
            loc.fixed64 = (op.flags & OSC_FIXED_PHASE) != 0;
        } else {
            loc.row = op.state_row + OSC_S_SYNC_LAST;
            loc.flag = true;
        }
        break;
    case SRACK_MOD_MOOG_FILTER:
        if (field <= SRACK_VCF_ST_Q)
            loc.row = op.state_row + VCF_S_F + (field - SRACK_VCF_ST_F);
        else if (field <= SRACK_VCF_ST_B4)
            loc.row = op.state_row + VCF_S_B0 + (field - SRACK_VCF_ST_B0);
        else
            loc.row = op.state_row + VCF_S_FREQ + (field - SRACK_VCF_ST_FREQ);
        break;
    case SRACK_MOD_GRID_SEQUENCER:
        loc.row = op.state_row + (field - SRACK_GRIDSEQ_CURRENT_STEP);
        loc.flag = field != SRACK_GRIDSEQ_LAST;
        break;
    case SRACK_MOD_PATTERN_SEQUENCER:
        loc.row = op.state_row + (field - SRACK_PATSEQ_CURRENT_STEP);
        loc.flag = true;
        break;
    case SRACK_MOD_SAMPLE:
        loc.row = op.state_row + (field - SRACK_SAMPLE_POS);
        loc.flag = field != SRACK_SAMPLE_POS;
        break;
    case SRACK_MOD_ADSR:
        switch (field) {
        case SRACK_ADSR_PHASE: loc.row = op.state_row + ADSR_S_PHASE; break;
        case SRACK_ADSR_MODE: loc.row = op.state_row + ADSR_S_MODE; loc.flag = true; break;
        case SRACK_ADSR_R_VAL: loc.row = op.state_row + ADSR_S_R_VAL; break;
        case SRACK_ADSR_FROM_A_VAL: loc.row = op.state_row + ADSR_S_FROM_A; break;
        case SRACK_ADSR_GATE_LAST: loc.row = op.state_row + ADSR_S_GATE_LAST; loc.flag = true; break;
        }
        break;
    }
    return loc;
}

int flatten(Graph& g, uint32_t n_voices, const std::vector<VoiceOverride>& overrides, uint32_t render_flags, FlatPair& out)
{
    out = FlatPair{};
    if (n_voices == 0) {
        set_error("flatten: n_voices = 0 (call srack_voices_configure first)");
        return SRACK_ERR_STATE;
    }
    if (!g.plan.valid) g.make_plan();
    const int n_mod = (int)g.modules.size();
    for (const auto& o : overrides) {
        if (o.module < 0 || o.module >= n_mod || o.field < 0 || o.field >= g.num_fields(o.module) || o.values.size() != n_voices) {
            set_error("flatten: bad per-voice override");
            return SRACK_ERR_INVALID;
        }
        int type = g.modules[(size_t)o.module].type;
        if ((type == SRACK_MOD_OSCILLATOR && o.field == SRACK_OSC_ANTIALIASING) || (type == SRACK_MOD_MATH && o.field == SRACK_MATH_OPERATION) ||
            (type == SRACK_MOD_GRID_SEQUENCER && (o.field == SRACK_GRIDSEQ_LENGTH || o.field == SRACK_GRIDSEQ_OCTAVES)) ||
            (type == SRACK_MOD_PATTERN_SEQUENCER && o.field == SRACK_PATSEQ_LENGTH) || (type == SRACK_MOD_SAMPLE && o.field == SRACK_SAMPLE_WAVE_NEW)) {
            set_error("flatten: antialiasing / operation / sequence length / wavebox.new are structural and cannot differ per voice");
            return SRACK_ERR_UNSUPPORTED;
        }
    }
    Analysis A;
    A.live.assign((size_t)n_mod, 0);
    A.port_live.assign((size_t)n_mod, 0);
    A.in_ctl.assign((size_t)n_mod, 0);
    out.in_ctl = A.in_ctl;
    const int output = g.plan.output;
    static const std::vector<VoiceOverride> kNoOverrides;
    if (output < 0) {  // no OutputModule: empty plan (ui.rs:75-79) => silence
        out.voice.hdr.n_channels = (int)g.cfg.channels;
        out.voice.hdr.buffer_size = (int)g.cfg.buffer_size;
        for (int c = 0; c < 8; c++) out.voice.hdr.channel_plane[c] = -1;
        out.voice.hdr.tile = kTileMax;
        out.voice.n_voices = n_voices;
        out.voice.op_of_module.assign((size_t)n_mod, -1);
        out.description = out.voice.description = "no OutputModule: empty plan";
        return SRACK_OK;
    }

    // ---- 1. reachability from the output, 2. live ports -------------------------------------------
    {
        int self_loop = -1;
        if (audible(g, A.live, A.port_live, &self_loop) != 0) {
            set_error("module " + std::to_string(self_loop) + " is wired to itself: the reference deadlocks on this (synth.rs:99,251)");
            return SRACK_ERR_SELF_LOOP;
        }
    }

    if (render_flags & kFlattenEvalAll) {  // everything the planner scheduled runs (plan_execution covers all_modules, synth.rs:107-212)
        for (int m : g.plan.order) {
            if (g.modules[(size_t)m].type == SRACK_MOD_OUTPUT && m != output) continue;  // (a second OutputModule is never observable)
            A.live[(size_t)m] = 1;
            for (const InputRef& in : g.modules[(size_t)m].in)
                if (in.src >= 0) {
                    if (in.src == m) {
                        set_error("module " + std::to_string(m) + " is wired to itself: the reference deadlocks on this (synth.rs:99,251)");
                        return SRACK_ERR_SELF_LOOP;
                    }
                    A.port_live[(size_t)in.src] |= 1u << in.port;
                }
        }
        render_flags &= ~kFlattenEvalAll;
    }

    // ---- 2b. which of the default mode's approximations this patch may take --------------------------------------------------------
    // (approx.cpp: a first-order error bound per wire — epsilon of each cheaper form times the gain from its wire to every output channel,
    // against half the contract; a patch with an unbounded gain behind something that has no exact form of its own goes exact altogether)
    {
        const ApproxPlan plan = plan_approximations(g, A.live, A.port_live, overrides, (render_flags & SRACK_RENDER_EXACT_OSC) != 0);
        A.sine_loose = plan.sine_loose;
        A.nonlin_loose = plan.nonlin_loose;
        A.saw_fixed = plan.saw_fixed;
        A.osc_exact = plan.osc_exact;
        A.exact_src.assign((size_t)n_mod, 0);
        for (int m = 0; m < n_mod; m++) A.exact_src[(size_t)m] = plan.exact_blep[(size_t)m] || plan.literal[(size_t)m];
        std::string oscs;  // the oscillators that are exact as a whole
        for (int m = 0; m < n_mod; m++)
            if (plan.osc_exact[(size_t)m]) oscs += (oscs.empty() ? "" : ",") + std::to_string(m);
        if (render_flags & SRACK_RENDER_EXACT_OSC) {
            // (asked for: nothing to decide)
        } else if ((render_flags & SRACK_RENDER_KEEP_DEFAULT) && (plan.exact_patch || !oscs.empty()) && !plan.unbounded_values) {
            // (KEEP_DEFAULT waives the unbounded-GAIN cases only: where the VALUES have no bound the default forms' v_med3 clamps send a NaN to -1
            // where the reference's min / max send it to +1 — wrong from the first overflow on, not "inside the contract for seconds")
            std::fill(A.osc_exact.begin(), A.osc_exact.end(), 0);
            out.approx_note = "kept default: " + (plan.exact_patch ? plan.why : "unbounded gain behind oscillator " + oscs);
        } else if (plan.exact_patch) {
            render_flags |= SRACK_RENDER_EXACT_OSC;
            std::fill(A.sine_loose.begin(), A.sine_loose.end(), 0);  // (the exact oscillator has one sine)
            out.approx_note = "exact: " + plan.why;
        } else {
            char buf[64];
            snprintf(buf, sizeof buf, "bound %.1e", plan.bound);
            out.approx_note = buf;
            if (!oscs.empty()) out.approx_note += "; exact osc " + oscs;
        }
    }
    out.effective_flags = render_flags;

    // ---- 3. voice-invariant sub-graph ----------------------------------------------------------------
    if (n_voices > 1 && !(render_flags & SRACK_RENDER_NO_UNIFORM_HOIST)) {
        std::vector<char>& u = A.in_ctl;
        for (int m = 0; m < n_mod; m++) u[(size_t)m] = A.live[(size_t)m] && m != output && g.modules[(size_t)m].type != SRACK_MOD_NOISE;  // every voice draws its own noise
        for (const auto& o : overrides) u[(size_t)o.module] = 0;
        const auto& pos = g.plan.position;
        for (bool changed = true; changed;) {
            changed = false;
            for (int m = 0; m < n_mod; m++) {
                if (!A.live[(size_t)m]) continue;
                for (const InputRef& in : g.modules[(size_t)m].in) {
                    if (in.src < 0) continue;
                    if (u[(size_t)m] && !u[(size_t)in.src]) {  // fed by something per-voice
                        u[(size_t)m] = 0;
                        changed = true;
                    }
                    // a per-voice module reading a uniform source through a broken (delayed) edge would need the
                    // track's history across renders: keep such a source per-voice instead
                    if (!u[(size_t)m] && u[(size_t)in.src] && pos[(size_t)in.src] > pos[(size_t)m]) {
                        u[(size_t)in.src] = 0;
                        changed = true;
                    }
                }
            }
        }
        for (int m : g.plan.order) {
            if (!A.live[(size_t)m] || u[(size_t)m]) continue;
            for (const InputRef& in : g.modules[(size_t)m].in)
                if (in.src >= 0 && u[(size_t)in.src] && !A.track_of.count({in.src, in.port})) {
                    A.track_of[{in.src, in.port}] = (int)A.tracks.size();
                    A.tracks.emplace_back(in.src, in.port);
                }
        }
        if (A.tracks.empty()) std::fill(u.begin(), u.end(), 0);
    }
    out.in_ctl = A.in_ctl;
    A.stage.assign((size_t)n_mod, -1);
    for (int m = 0; m < n_mod; m++)
        if (A.in_ctl[(size_t)m]) A.stage[(size_t)m] = 0;

    // ---- 3b. control units ---------------------------------------------------------------------------------
    // One wave evaluating one voice is a latency chain: ~0.1-0.3 us per module per sample, and the voice kernels of
    // a chunk cannot start before the chunk's tracks exist.  Control modules only exchange finished samples, so a
    // module at dependency depth d can work on chunk k while its consumers work on chunk k-1: every module becomes
    // its own unit (one block of the control launch), trailing by its depth; every wire between units is a track.
    // The launch then lasts as long as its slowest module, not as the sum.  A feedback edge inside the control
    // sub-graph would tie depths together, and a small control program is not worth the extra tracks (it may also be
    // the fused gate -> envelope shape): both stay one unit.
    constexpr int kMinModulesToSplit = 4;
    std::vector<int> unit_lag{0};
    if (!A.tracks.empty() && !(render_flags & SRACK_RENDER_NO_CTL_STAGES)) {
        const auto& pos = g.plan.position;
        int n_ctl = 0;
        bool feedback = false;
        for (int m = 0; m < n_mod; m++) {
            if (!A.in_ctl[(size_t)m]) continue;
            n_ctl++;
            for (const InputRef& in : g.modules[(size_t)m].in)
                if (in.src >= 0 && A.in_ctl[(size_t)in.src] && pos[(size_t)in.src] > pos[(size_t)m]) feedback = true;
        }
        if (n_ctl >= kMinModulesToSplit && feedback && !getenv("SRACK_CTL_ONE_UNIT_ON_FEEDBACK")) {
            // Feedback inside the control sub-graph: the modules of a cycle must share a unit (the delayed edge is a ring inside it), everything
            // else is pipelined as without feedback — units are the strongly connected components of the control sub-graph (Tarjan; a wire is
            // an edge whether the planner delayed it or not).  Until round 3 one such edge kept the WHOLE control program one unit: one lane's
            // latency chain through all its modules, 0.6 - 1.4 us per sample on random patches (NOTES R3.10).
            std::vector<int> index((size_t)n_mod, -1), low((size_t)n_mod, 0), comp((size_t)n_mod, -1), stack;
            std::vector<char> on_stack((size_t)n_mod, 0);
            int counter = 0, n_comp = 0;
            std::vector<std::vector<int>> sinks_of((size_t)n_mod);
            for (int m = 0; m < n_mod; m++)
                if (A.in_ctl[(size_t)m])
                    for (const InputRef& in : g.modules[(size_t)m].in)
                        if (in.src >= 0 && A.in_ctl[(size_t)in.src]) sinks_of[(size_t)in.src].push_back(m);
            struct Frame { int v; size_t next; };
            for (int root = 0; root < n_mod; root++) {
                if (!A.in_ctl[(size_t)root] || index[(size_t)root] >= 0) continue;
                std::vector<Frame> call{{root, 0}};
                index[(size_t)root] = low[(size_t)root] = counter++;
                stack.push_back(root);
                on_stack[(size_t)root] = 1;
                while (!call.empty()) {
                    Frame& f = call.back();
                    if (f.next < sinks_of[(size_t)f.v].size()) {
                        const int w = sinks_of[(size_t)f.v][f.next++];
                        if (index[(size_t)w] < 0) {
                            index[(size_t)w] = low[(size_t)w] = counter++;
                            stack.push_back(w);
                            on_stack[(size_t)w] = 1;
                            call.push_back({w, 0});
                        } else if (on_stack[(size_t)w]) {
                            low[(size_t)f.v] = std::min(low[(size_t)f.v], index[(size_t)w]);
                        }
                    } else {
                        const int v = f.v;
                        if (low[(size_t)v] == index[(size_t)v]) {
                            for (;;) {
                                const int w = stack.back();
                                stack.pop_back();
                                on_stack[(size_t)w] = 0;
                                comp[(size_t)w] = n_comp;
                                if (w == v) break;
                            }
                            n_comp++;
                        }
                        call.pop_back();
                        if (!call.empty()) low[(size_t)call.back().v] = std::min(low[(size_t)call.back().v], low[(size_t)v]);
                    }
                }
            }
            if (n_comp >= 2) {  // (one component: the whole control program is one cycle — nothing to pipeline)
                std::vector<int> unit_of_comp((size_t)n_comp, -1);
                unit_lag.clear();
                for (int m : g.plan.order) {  // units numbered by first appearance in the plan
                    if (!A.in_ctl[(size_t)m]) continue;
                    int& u = unit_of_comp[(size_t)comp[(size_t)m]];
                    if (u < 0) {
                        u = (int)unit_lag.size();
                        unit_lag.push_back(0);
                    }
                    A.stage[(size_t)m] = u;
                }
                for (bool changed = true; changed;) {  // depths over the condensation (a DAG): a unit trails every unit it reads by one chunk
                    changed = false;
                    for (int m = 0; m < n_mod; m++) {
                        if (!A.in_ctl[(size_t)m]) continue;
                        for (const InputRef& in : g.modules[(size_t)m].in) {
                            if (in.src < 0 || !A.in_ctl[(size_t)in.src] || A.stage[(size_t)in.src] == A.stage[(size_t)m]) continue;
                            const int need = unit_lag[(size_t)A.stage[(size_t)in.src]] + 1;
                            if (unit_lag[(size_t)A.stage[(size_t)m]] < need) {
                                unit_lag[(size_t)A.stage[(size_t)m]] = need;
                                changed = true;
                            }
                        }
                    }
                }
            }
        } else if (n_ctl >= kMinModulesToSplit && !feedback) {
            unit_lag.clear();
            for (int m : g.plan.order) {  // plan order: every non-delayed source comes first
                if (!A.in_ctl[(size_t)m]) continue;
                int depth = 0;
                for (const InputRef& in : g.modules[(size_t)m].in)
                    if (in.src >= 0 && A.in_ctl[(size_t)in.src]) depth = std::max(depth, unit_lag[(size_t)A.stage[(size_t)in.src]] + 1);
                A.stage[(size_t)m] = (int)unit_lag.size();
                unit_lag.push_back(depth);
            }
        }
        if (unit_lag.size() > 1) {
            for (int m : g.plan.order) {  // wires between units travel as tracks too
                if (!A.in_ctl[(size_t)m]) continue;
                for (const InputRef& in : g.modules[(size_t)m].in)
                    if (in.src >= 0 && A.stage[(size_t)in.src] != A.stage[(size_t)m] && !A.track_of.count({in.src, in.port})) {
                        A.track_of[{in.src, in.port}] = (int)A.tracks.size();
                        A.tracks.emplace_back(in.src, in.port);
                    }
            }
        }
    }
    const int n_stages = (int)unit_lag.size();
    out.ctl_stage = A.stage;
    out.ctl_lag = unit_lag;
    out.n_tracks = (int)A.tracks.size();

    if (out.n_tracks > 0) {
        out.ctl.resize((size_t)n_stages);
        for (int st = 0; st < n_stages; st++) {
            Builder bc{g, 1, kNoOverrides, render_flags, A, st, true, out.ctl[(size_t)st], {}, {}};
            int rc = bc.build();
            if (rc != SRACK_OK) return rc;
        }
    }
    Builder bv{g, n_voices, overrides, render_flags, A, -1, false, out.voice, {}, {}};
    int rc = bv.build();
    if (rc != SRACK_OK) return rc;
    std::ostringstream d;
    d << out.voice.description;
    for (const FlatProgram& c : out.ctl) d << " + " << c.description;
    if (out.n_tracks > 0) d << " tracks=" << out.n_tracks;
    d << " B=" << g.cfg.buffer_size;
    if (!out.approx_note.empty()) d << " approx[" << out.approx_note << "]";
    out.description = d.str();
    return SRACK_OK;
}

}  // namespace srack
