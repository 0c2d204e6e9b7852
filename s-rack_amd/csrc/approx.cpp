// approx.cpp — the default mode's approximations as an error budget (see approx.hpp).
//
//   1. magnitudes: sup |value| per wire, forward (|sine| <= 1, |saw|, |square| <= 2, |lowpass| <= 1, sums and products through the arithmetic)
//   2. per filter: the L1 norms of the ladder's small-signal impulse responses over the cutoffs it can reach (the exact L-infinity gain of
//      a linear filter; the cubic and the clamps only compress), its sensitivity to the cutoff, and how its cutoff moves
//   3. gains: per output channel, backward: G(wire) = sum over the inputs that read it of sum over that module's outputs of
//      g(input -> output) * G(output wire); a fixpoint over the cycles — converging to 1 / (1 - loop gain) or declared unbounded
//   4. decisions: every approximated form is an epsilon on its wire; forms are denied, largest contribution first, until the sum of
//      epsilon * G stays below kApproxBudget on every channel; an oscillator behind an unbounded gain whose pitch moves, or whose sine is
//      heard there, is evaluated exactly as a whole (OSC_EXACT on that op); values without a bound, or an unbounded gain behind the sample
//      player's pitch (no exact form of its own), turn the whole patch exact.
//
// Where the numbers come from.  Epsilons: the f32 sine — two roundings at values in [1, 2): 2.4e-7; the f32 PolyBLEP — measured on the kernels' forms: 3.2e-7 (kEpsBlep).  The contracted ladder —
// tools/ladder_calib.c emulates both forms on the CPU (filter.rs:58-92 against modules.hip.h vcf_step<true>): over resonance 0 ... 0.89,
// cutoff 0.02 ... 0.9, saw inputs and still / ramped / sine-swept cutoffs the difference stays below 1.3e-6 (lowpass), 3.9e-6 (bandpass),
// 3.3e-6 (highpass) whatever the L1 norm (the roundings are not aligned with the impulse response); a cutoff that JUMPS at audio rate (a
// square, noise on the CV) breaks that: 5e-5 and worse — such a filter has no contracted form.  Gains: the L1 norms are computed here, per
// filter, from its coefficients (resonance 0.5: 1.1 ... 3.2; 0.89: up to 47 at cutoff 0.2; from ~0.9 the linear ladder does not decay at
// mid cutoffs: unbounded); the sensitivity to the cutoff is measured at up to 2.7 / cutoff times the port's L1 norm (same tool: 3.0 here).
// What is NOT bounded here: the default forms' own last-bit differences from the reference's libm (polynomial 2^cv at 3e-16, 2e-14 inside
// the proved bounded-CV classes — 1e-10 cycles of phase per minute; the degree-8 series they had through round 5 left 1.2e-8 and showed on the
// minute's curve —; the f64 sine at one rounding), measured over minutes instead (profiles/r06_horizon.json);
// they only decide anything where a gain is unbounded: there the oscillator takes the reference's own forms.
#include "approx.hpp"

#include <algorithm>
#include <cmath>
#include <limits>

namespace srack {

namespace {

constexpr double kInf = std::numeric_limits<double>::infinity();
constexpr double kBig = 1e200;              // a gain or magnitude beyond this counts as unbounded (chains of event and pitch gains stay far below; a cycle that multiplies gets there)
constexpr double kEventGain = 1e9;          // an event input (`value > 0.0` decides when something happens: a gate, a sync, a step): an error e moves an
                                            // edge by a sample wherever |value| < e at a crossing, and a moved edge is an error of O(1) — the "gain" is
                                            // 1 / (the |value| a crossing may be trusted at): forms with e > 5e-15 are denied in front of an event
constexpr double kEpsBlep = 3.2e-7;         // f32 PolyBLEP against the f64 one, MEASURED on the kernels' own forms (tools/blep_calib.py, 2 048 pitches x 1 s, through tests/cpp/
                                            // forms_emu.c, which tools/emu_vs_gpu.py holds to the kernels bit for bit): osc_step's 1.2e-7, the carried-phase saw 2.42e-7, the
                                            // fixed-point saw 3.15e-7 on top of its window term (kEpsFixedWindow) — with 1 / dt rounded once (modules.hip.h, inv_dt_f32)
constexpr double kEpsSine = 2.4e-7;         // f32 sine after the exact f64 fold
constexpr double kEpsFixed = 3.2e-12;       // 2^-64 fixed-point phase: kApproxHorizon steps x 2^-64 = 1.6e-12 of phase, saw slope 2 ...
constexpr double kEpsFixedWindow = 0x1p-31; // ... PLUS, inside the two PolyBLEP windows, t = pos / dt taken from the phase's UPPER 32 bits (modules.hip.h, fosc_saw: c32 = f32(hi)):
                                            // pos is below dt there, so the truncation to 2^-32 is an error of 2^-32 / dt in t and of up to twice that in -(1 - t)^2 —
                                            // nothing at 440 Hz (5e-8), 1.0e-6 at 17 Hz (round 6, tools/emu_vs_gpu.py: seed 900146 rendered 1.02e-6 where the bound said
                                            // 2.4e-7), 2.5e-5 for a 0.9 Hz LFO: the form's epsilon is this over the SMALLEST increment any voice has
constexpr double kEpsNonlin = 4e-6;         // v_log_f32 / v_exp_f32 power, relative to max(|out|, 1)
                                            // (the power WITHOUT that form is the host libm's powf operation for operation since round 6 — modules.hip.h,
                                            // powf_libm_plain — and costs nothing here; until then it was an f32 ulp away now and then, which ADVICE r05
                                            // found counted as 0 behind unbounded gains)
constexpr double kEpsLadder[3] = {1.5e-6, 4.2e-6, 3.6e-6};  // contracted ladder, lowpass / bandpass / highpass (tools/ladder_calib.c)
constexpr double kLadderNoiseInput = 1.5;   // ... with a noise-like signal on the audio input (noise, a sample player, a reverb: a new level every sample excites the resonance all the
                                            // time, a saw now and then): `noisein`, 2.0 / 5.4 / 1.7e-6 up to resonance 0.6 — and 8.5e-6 / 1.2e-5 / 6.1e-6 from 0.8 up (0.7 with
                                            // an LFO on the cutoff): no contracted form above kLadderNoiseInputRes (tools/cpu_soak.py, noise family, seeds 235484 ...)
constexpr double kLadderNoiseInputRes = 0.6;
constexpr double kLadderRareJumps = 2.0;    // ... with a cutoff that jumps now and then (an envelope's attack, a sequencer's step): same tool, 7e-6 on the bandpass
constexpr double kLadderDriveMax = 1.75;    // above this input amplitude a ladder is OVERDRIVEN: its stages sit in their clamps and flip within a sample, and the last stage's
                                            // cubic, b4 - b4^3 / 6 — slope 1 - b4^2 / 2, below -1 past |b4| = 2 — with the stage's own feedback - b4 * f around it is an
                                            // expanding map: chaos without any resonance.  tools/ladder_calib.c `amp`: the literal ladder's response to a 2.4e-7 disturbance of
                                            // its input is 10 - 29 x up to an amplitude of 1.75, 1e4 - 1e6 x from 1.9 up at low resonance (round 5's seeds 105055, 123042)
constexpr double kLadderTameCutoff = 0.35;  // ... but only where the cutoff passes this: below, the stage's coefficient p keeps the last stage's value under the cubic's turning
                                            // point whatever the drive (same tool, `tame`: <= 23 x at cutoffs <= 0.4 for amplitudes up to 1000; 35 x at 0.42 and 3 700 x at 0.46 for 8 - 16)
constexpr double kLadderL1Max = 64.0;       // beyond this lowpass L1 norm the ladder is treated as self-oscillating (the calibration stops at 47)
constexpr double kNonlinSteep = 1e4;        // d|a|^b / da near a = 0 for b < 1: (2.4e-7)^0.5 / 2.4e-7 = 2e3
constexpr int kSweeps = 400;

enum : uint32_t { kJumpAudio = 1u, kJumpRare = 2u, kJumpNoise = 4u };  // how a wire moves, for a cutoff CV: edges at audio rate (a square, a saw), now and then (an
                                                                          // envelope's attack, a sequencer's step), a new level every sample (noise, a sample player)

struct Range {
    double lo = 0.0, hi = 0.0;
    double abs_max() const { return std::max(std::fabs(lo), std::fabs(hi)); }
};

struct Ladder {
    double l1[3] = {0.0, 0.0, 0.0};      // L1 norm of the impulse response audio -> lowpass / bandpass / highpass, worst over the reachable cutoffs
    double cutoff[3] = {0.0, 0.0, 0.0};  // gain cutoff CV -> port
    bool stable = true;
    double own = 1.0;                    // factor on kEpsLadder for the ladder's own rounding (contracted, or literal behind a perturbed input): inf = none claimed
    bool overdriven = false;             // the audio input can exceed kLadderDriveMax (no contracted form; unbounded where the cutoff passes kLadderTameCutoff)
    uint32_t motion = 0;                 // kJump* of the cutoff CV
};

// L1 norms of the small-signal ladder (filter.rs:61-82 without the cubic and the clamps) at one (cutoff, resonance); false: it does not decay
bool ladder_l1(double fr, double res, double out[3])
{
    const double q0 = 1.0 - fr, p = fr + 0.8 * fr * q0, f = 2.0 * p - 1.0, q = res * (1.0 + 0.5 * q0 * (1.0 - q0 + 5.6 * q0 * q0));
    double b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0, b4 = 0.0, window = 0.0;
    out[0] = out[1] = out[2] = 0.0;
    constexpr int kMax = 1 << 16, kWindow = 512;
    for (int i = 0; i < kMax; i++) {
        const double in = (i == 0 ? 1.0 : 0.0) - q * b4;
        double t1 = b1;
        b1 = (in + b0) * p - b1 * f;
        const double t2 = b2;
        b2 = (b1 + t1) * p - b2 * f;
        t1 = b3;
        b3 = (b2 + t2) * p - b3 * f;
        b4 = (b3 + t1) * p - b4 * f;
        b0 = in;
        const double a = std::fabs(b4) + std::fabs(b3) + std::fabs(in);
        out[0] += std::fabs(b4);
        out[1] += std::fabs(3.0 * (b3 - b4));
        out[2] += std::fabs(in - b4);
        window += a;
        if (!(a < 1e6)) return false;
        if ((i + 1) % kWindow == 0) {
            if (window < 1e-9) return true;
            window = 0.0;
        }
    }
    return false;
}

struct Analysis {
    const Graph& g;
    const std::vector<char>& live;
    const std::vector<uint32_t>& port_live;
    const std::vector<VoiceOverride>& ov;
    const int n_mod;
    const int output;
    const double sr;
    std::vector<std::vector<double>> mag;
    std::vector<std::vector<uint32_t>> motion;
    std::vector<int> scc;                 // strongly connected component per live module (-1: not live)
    std::vector<char> on_cycle;
    std::vector<Ladder> ladder;           // per module (filters only)
    struct Reader { int k, i; };
    std::vector<std::vector<std::vector<Reader>>> readers;  // [module][port] -> inputs that read the wire

    Analysis(const Graph& g_, const std::vector<char>& live_, const std::vector<uint32_t>& pl, const std::vector<VoiceOverride>& ov_)
        : g(g_), live(live_), port_live(pl), ov(ov_), n_mod((int)g_.modules.size()), output(g_.plan.output), sr((double)g_.cfg.sample_rate)
    {
        readers.resize((size_t)n_mod);
        for (int m = 0; m < n_mod; m++) readers[(size_t)m].resize((size_t)std::max(g.modules[(size_t)m].n_out, 0));
        for (int k = 0; k < n_mod; k++) {
            if (!live[(size_t)k]) continue;
            const Module& sink = g.modules[(size_t)k];
            for (int i = 0; i < sink.n_in; i++) {
                const InputRef& in = sink.in[(size_t)i];
                if (in.src >= 0 && live[(size_t)in.src] && in.port < g.modules[(size_t)in.src].n_out) readers[(size_t)in.src][(size_t)in.port].push_back({k, i});
            }
        }
    }

    int type(int m) const { return g.modules[(size_t)m].type; }
    bool port_is_live(int m, int p) const { return (port_live[(size_t)m] >> p) & 1u; }

    mutable std::vector<std::vector<std::pair<char, Range>>> range_cache;  // [module][field]: (computed, range) — an override holds one value per voice
    Range field(int module, int f) const
    {
        if (range_cache.empty()) range_cache.resize((size_t)n_mod);
        auto& per = range_cache[(size_t)module];
        if (per.size() <= (size_t)f) per.resize((size_t)f + 1, {0, Range{}});
        if (!per[(size_t)f].first) per[(size_t)f] = {1, field_scan(module, f)};
        return per[(size_t)f].second;
    }
    Range field_scan(int module, int f) const
    {
        const VoiceOverride* hit = nullptr;
        for (const auto& o : ov)
            if (o.module == module && o.field == f) hit = &o;  // the last one wins (flatten.cpp, find_override)
        Range r;
        if (!hit || hit->values.empty()) {
            r.lo = r.hi = (double)(float)g.modules[(size_t)module].fields[(size_t)f];
            return r;
        }
        r.lo = kInf, r.hi = -kInf;
        for (double v : hit->values) {
            const double x = (double)(float)v;
            if (!(std::fabs(x) < kInf)) {
                r.lo = -kInf, r.hi = kInf;
                return r;
            }
            r.lo = std::min(r.lo, x);
            r.hi = std::max(r.hi, x);
        }
        return r;
    }

    double in_mag(int k, int i) const  // sup |value| an input reads (0: unconnected)
    {
        const InputRef& in = g.modules[(size_t)k].in[(size_t)i];
        if (in.src < 0 || !live[(size_t)in.src] || in.port >= (int)mag[(size_t)in.src].size()) return 0.0;
        return mag[(size_t)in.src][(size_t)in.port];
    }
    uint32_t in_motion(int k, int i) const
    {
        const InputRef& in = g.modules[(size_t)k].in[(size_t)i];
        if (in.src < 0 || !live[(size_t)in.src] || in.port >= (int)motion[(size_t)in.src].size()) return 0u;
        return motion[(size_t)in.src][(size_t)in.port];
    }
    bool connected(int k, int i) const { return g.modules[(size_t)k].in[(size_t)i].src >= 0; }

    double osc_delta_max(int k) const  // sup of the phase increment, oscillator.rs:43-48
    {
        const double cv = connected(k, SRACK_OSC_IN_CV) ? in_mag(k, SRACK_OSC_IN_CV) : 0.0;
        if (!(cv < 64.0)) return kInf;
        return 440.0 * std::exp2(field(k, SRACK_OSC_VAL).hi + cv) / sr;
    }

    Range nonlin_exponent(int k) const
    {
        if (connected(k, 1)) {
            const double m = in_mag(k, 1);
            return Range{-m, m};
        }
        return field(k, SRACK_NONLIN_CONSTANT);
    }

    // ---- strongly connected components of the live graph (Tarjan, iterative; a wire is an edge whether the planner delayed it or not) ----
    void components()
    {
        scc.assign((size_t)n_mod, -1);
        on_cycle.assign((size_t)n_mod, 0);
        std::vector<int> index((size_t)n_mod, -1), low((size_t)n_mod, 0), stack;
        std::vector<char> on_stack((size_t)n_mod, 0);
        std::vector<std::vector<int>> succ((size_t)n_mod);
        for (int m = 0; m < n_mod; m++)
            if (live[(size_t)m])
                for (const auto& port : readers[(size_t)m])
                    for (const Reader& r : port) succ[(size_t)m].push_back(r.k);
        int counter = 0, n_comp = 0;
        struct Frame { int v; size_t next; };
        std::vector<int> size;
        for (int root = 0; root < n_mod; root++) {
            if (!live[(size_t)root] || index[(size_t)root] >= 0) continue;
            std::vector<Frame> call{{root, 0}};
            index[(size_t)root] = low[(size_t)root] = counter++;
            stack.push_back(root);
            on_stack[(size_t)root] = 1;
            while (!call.empty()) {
                Frame& fr = call.back();
                if (fr.next < succ[(size_t)fr.v].size()) {
                    const int w = succ[(size_t)fr.v][fr.next++];
                    if (index[(size_t)w] < 0) {
                        index[(size_t)w] = low[(size_t)w] = counter++;
                        stack.push_back(w);
                        on_stack[(size_t)w] = 1;
                        call.push_back({w, 0});
                    } else if (on_stack[(size_t)w]) {
                        low[(size_t)fr.v] = std::min(low[(size_t)fr.v], index[(size_t)w]);
                    }
                } else {
                    const int v = fr.v;
                    if (low[(size_t)v] == index[(size_t)v]) {
                        int count = 0;
                        for (;;) {
                            const int w = stack.back();
                            stack.pop_back();
                            on_stack[(size_t)w] = 0;
                            scc[(size_t)w] = n_comp;
                            count++;
                            if (w == v) break;
                        }
                        size.push_back(count);
                        n_comp++;
                    }
                    call.pop_back();
                    if (!call.empty()) low[(size_t)call.back().v] = std::min(low[(size_t)call.back().v], low[(size_t)v]);
                }
            }
        }
        for (int m = 0; m < n_mod; m++)
            if (live[(size_t)m]) on_cycle[(size_t)m] = size[(size_t)scc[(size_t)m]] > 1;
    }
    bool same_cycle(int a, int b) const { return on_cycle[(size_t)a] && scc[(size_t)a] == scc[(size_t)b]; }

    // ---- 1. magnitudes and how a wire moves --------------------------------------------------------------------------------------
    void out_mag(int m, std::vector<double>& o, std::vector<uint32_t>& mv) const
    {
        const Module& mod = g.modules[(size_t)m];
        auto any_motion = [&]() {
            uint32_t u = 0;
            for (int i = 0; i < mod.n_in; i++) u |= in_motion(m, i);
            return u;
        };
        auto clock_motion = [&](int k) -> uint32_t {
            return ((in_motion(k, SRACK_SEQ_IN_STEP) | in_motion(k, SRACK_SEQ_IN_SYNC)) & (kJumpAudio | kJumpNoise)) ? (uint32_t)kJumpAudio : (uint32_t)kJumpRare;
        };
        switch (mod.type) {
        case SRACK_MOD_OSCILLATOR: {
            // the square stays within +-1 at any increment, the saw (band-limited or not) up to an increment of one cycle per sample (beyond,
            // every sample falls into the PolyBLEP window and the correction adds up to 1: 2 pos)
            const double edge_mag = osc_delta_max(m) <= 1.0 ? 1.0 : 2.0;
            o = {1.0, edge_mag, edge_mag};
            // below 48 Hz: an LFO's edges — unless something hard-syncs it: the resets come at the sync source's rate, and each is a raw jump of
            // the saw (round 5's soak at 200 voices x 6 000 samples, seed 66697: a 22 Hz saw, synced by a filter's highpass, on a second filter's
            // cutoff: 4.5e-5 in the contracted form of that filter, in 8 voices of 200)
            const uint32_t edges = osc_delta_max(m) < 1e-3 && !connected(m, SRACK_OSC_IN_SYNC) ? kJumpRare : kJumpAudio;
            // (the sine of an oscillator above LFO rate has no edges, but it moves by a good part of its range from one sample to the next:
            // for a cutoff that is the same thing — the calibration's "smooth" cutoffs are envelopes, LFOs and sines up to 1.8 kHz)
            // ... and RAW jumps at audio rate — a hard sync's resets, the edges of an oscillator without anti-aliasing — excite a filter's resonance
            // like noise does, not like the band-limited saw the ladder's figures were measured with (tools/cpu_soak.py, seeds 226856 and 405576:
            // hard-synced saws into contracted ladders at resonance 0.79 / 0.91, 5.2e-6 where the bandpass's figure is 4.2e-6, 1.9e-6 for 1.5e-6)
            const uint32_t raw = edges == kJumpAudio && (connected(m, SRACK_OSC_IN_SYNC) || field(m, SRACK_OSC_ANTIALIASING).lo == 0.0) ? (uint32_t)kJumpNoise : 0u;
            mv = {(edges == kJumpAudio ? (uint32_t)kJumpAudio : 0u) | (connected(m, SRACK_OSC_IN_SYNC) ? raw : 0u), edges | raw, edges | raw};
            break;
        }
        case SRACK_MOD_MOOG_FILTER: {
            const double res = std::min(std::max(field(m, SRACK_VCF_RES).hi, 0.0), 1.0);
            o = {1.0, 6.0, in_mag(m, SRACK_VCF_IN_AUDIO) + 3.8 * res + 1.0};
            // every port keeps the input's edges: band- and highpass by construction, and a lowpass only smooths them at a low cutoff — at 0.9
            // a square comes out a square (round 5's soak through the specialised kernels, seed 72223: one filter's lowpass, fed from a
            // chaotic loop, on a second filter's cutoff — 5.3e-5 on that one's contracted highpass)
            // ... and a filter whose own cutoff jumps hands that on, input or no input (seed 104123, FUZZ_MORE_OV at 200 voices: a ladder past
            // self-oscillation with nothing on its audio input and a square on its cutoff, its highpass on the next filter's cutoff: 5.2e-4 there)
            const uint32_t a = in_motion(m, SRACK_VCF_IN_AUDIO) | (connected(m, SRACK_VCF_IN_CV) ? in_motion(m, SRACK_VCF_IN_CV) : 0u);
            mv = {a, a, a};
            break;
        }
        case SRACK_MOD_ADSR:
            o = {std::max(1.0, field(m, SRACK_ADSR_S_VAL).abs_max())};
            // an envelope moves now and then — as often as its gate opens: gated by something that moves at audio rate (an oscillator above LFO
            // rate, noise) it restarts every few samples (tools/cpu_soak.py, noise family, seed 277445: white noise on an envelope's gate, the
            // envelope on a cutoff: the contracted lowpass at 1.0e-5 where the bound said 3e-6)
            mv = {(in_motion(m, 0) & (kJumpAudio | kJumpNoise)) ? (uint32_t)kJumpAudio : (uint32_t)kJumpRare};
            break;
        case SRACK_MOD_VCA:
            o = {connected(m, 0) && connected(m, 1) ? in_mag(m, 0) * in_mag(m, 1) : 0.0};
            mv = {any_motion()};
            break;
        case SRACK_MOD_MONO_MIXER: {
            double s = 0.0;
            for (int i = 0; i < 4; i++)
                if (connected(m, i)) s += field(m, SRACK_MIX_GAIN0 + i).abs_max() * in_mag(m, i);
            o = {s};
            mv = {any_motion()};
            break;
        }
        case SRACK_MOD_MATH: {
            const double a = connected(m, 0) ? in_mag(m, 0) : 0.0, b = connected(m, 1) ? in_mag(m, 1) : field(m, SRACK_MATH_CONSTANT).abs_max();
            o = {(int)mod.fields[SRACK_MATH_OPERATION] == SRACK_MATH_MULTIPLY ? (a == 0.0 || b == 0.0 ? 0.0 : a * b) : a + b};
            mv = {any_motion()};
            break;
        }
        case SRACK_MOD_NONLINEAR: {
            const double a = connected(m, 0) ? in_mag(m, 0) : 0.0;
            const Range b = nonlin_exponent(m);
            o = {b.lo < 0.0 ? kInf : std::max(std::pow(a, b.lo), std::pow(a, b.hi))};
            mv = {any_motion()};
            break;
        }
        case SRACK_MOD_SAMPLE: {
            double w = 0.0;
            for (float x : mod.wave) w = std::max(w, (double)std::fabs(x));
            o = {std::isfinite(w) ? w : kInf};
            mv = {(uint32_t)(kJumpAudio | kJumpNoise)};  // (whatever was recorded)
            break;
        }
        case SRACK_MOD_NOISE:
            o = {1.0};
            mv = {(uint32_t)(kJumpAudio | kJumpNoise)};
            break;
        case SRACK_MOD_FREEVERB: {  // 8 recirculating combs and 4 allpasses per channel: the bound of edge_of() times the louder input
            const double in = std::max(connected(m, 0) ? in_mag(m, 0) : 0.0, connected(m, 1) ? in_mag(m, 1) : 0.0);
            const double fb = std::min(std::fabs(mod.fields[SRACK_FREEVERB_ROOM_SIZE]) * 0.28 + 0.7, 0.999);
            const double gain = mod.fields[SRACK_FREEVERB_FREEZE] != 0.0 ? kInf : std::fabs(mod.fields[SRACK_FREEVERB_DRY]) + 3.0 * std::fabs(mod.fields[SRACK_FREEVERB_WET]) * 8.0 / (1.0 - fb) * 40.0;
            o = {in == 0.0 ? 0.0 : in * gain, in == 0.0 ? 0.0 : in * gain};
            mv = {any_motion(), any_motion()};
            break;
        }
        case SRACK_MOD_GRID_SEQUENCER: {
            double note = std::fabs(mod.fields[SRACK_GRIDSEQ_LAST]);
            const double spo = std::max(1.0, field(m, SRACK_GRIDSEQ_STEPS_PER_OCTAVE).lo);
            for (uint32_t c : mod.cells)
                if (c >> 31) note = std::max(note, (double)(c & 0xffffu) / spo);
            const double gate = std::max(1.0, connected(m, SRACK_SEQ_IN_STEP) ? in_mag(m, SRACK_SEQ_IN_STEP) : 0.0);
            o = {note, gate, 1.0};
            const uint32_t clocked = clock_motion(m);  // (a sequencer steps as often as its clock ticks)
            mv = {clocked, clocked | in_motion(m, SRACK_SEQ_IN_STEP), clocked};
            break;
        }
        case SRACK_MOD_PATTERN_SEQUENCER: {
            const double gate = std::max(1.0, connected(m, SRACK_SEQ_IN_STEP) ? in_mag(m, SRACK_SEQ_IN_STEP) : 0.0);
            o.assign((size_t)mod.n_out, gate);
            mv.assign((size_t)mod.n_out, kJumpRare | in_motion(m, SRACK_SEQ_IN_STEP));
            break;
        }
        default:
            o.assign((size_t)std::max(mod.n_out, 0), 0.0);
            mv.assign((size_t)std::max(mod.n_out, 0), 0u);
            break;
        }
        o.resize((size_t)std::max(mod.n_out, 0), 0.0);
        mv.resize((size_t)std::max(mod.n_out, 0), 0u);
        for (double& x : o)
            if (!(x < kBig)) x = kInf;
    }

    void magnitudes()
    {
        mag.assign((size_t)n_mod, {});
        motion.assign((size_t)n_mod, {});
        for (int m = 0; m < n_mod; m++) {
            mag[(size_t)m].assign((size_t)std::max(g.modules[(size_t)m].n_out, 0), 0.0);
            motion[(size_t)m].assign((size_t)std::max(g.modules[(size_t)m].n_out, 0), 0u);
        }
        std::vector<double> o;
        std::vector<uint32_t> mv;
        for (int sweep = 0; sweep < 64; sweep++) {
            bool changed = false;
            for (int m : g.plan.order) {
                if (!live[(size_t)m] || m == output) continue;
                out_mag(m, o, mv);
                for (size_t p = 0; p < o.size(); p++) {
                    if (sweep >= 48 && o[p] > mag[(size_t)m][p]) o[p] = kInf;  // a cycle that keeps growing
                    if (o[p] > mag[(size_t)m][p] * (1.0 + 1e-9) || mv[p] != motion[(size_t)m][p]) changed = true;
                    mag[(size_t)m][p] = std::max(mag[(size_t)m][p], o[p]);
                    motion[(size_t)m][p] |= mv[p];
                }
            }
            if (!changed) break;
        }
    }

    // ---- 2. the ladders -------------------------------------------------------------------------------------------------------------
    void ladders()
    {
        ladder.assign((size_t)n_mod, Ladder{});
        for (int m = 0; m < n_mod; m++) {
            if (!live[(size_t)m] || type(m) != SRACK_MOD_MOOG_FILTER) continue;
            Ladder& L = ladder[(size_t)m];
            const Range freq = field(m, SRACK_VCF_FREQ), amount = field(m, SRACK_VCF_EXP_AMT);
            const double res = std::min(std::max(field(m, SRACK_VCF_RES).hi, 0.0), 1.0);  // filter.rs:214
            const bool has_cv = connected(m, SRACK_VCF_IN_CV);
            const double swing = has_cv ? in_mag(m, SRACK_VCF_IN_CV) * amount.abs_max() : 0.0;
            double lo = freq.lo - swing, hi = freq.hi + swing;  // filter.rs:213: (freq + cv * exp_amt).max(0.0).min(0.9)
            if (!(lo > 0.0)) lo = 0.0;
            if (!(hi < 0.9)) hi = 0.9;
            lo = std::min(lo, 0.9), hi = std::max(hi, 0.0);
            if (lo > hi) std::swap(lo, hi);
            L.motion = has_cv ? in_motion(m, SRACK_VCF_IN_CV) : 0u;
            // A cutoff that moves at audio rate makes the ladder a time-varying system, and the static norms no longer bound it.  tools/ladder_calib.c
            // (`gain`: the literal ladder's response to a 2.4e-7 disturbance of its input): with a square on the CV the response stays within 3.5 x
            // the static one up to resonance 0.8 for a unit saw on the input and is unbounded above; with white noise on the CV it is unbounded from
            // resonance 0.2 up.  And round 5's soak with per-voice parameters at 200 voices (seed 105055) has a literal ladder at resonance 0.52
            // whose cutoff an audio-rate saw moves between 0.44 and the clamp at 0.9, with an input of magnitude 4 driving it into its clamps:
            // a 4.8e-7 disturbance on the input came out at 6.9e-5 — 145 x, eleven times the static norm times four.  No bound is claimed for
            // such a filter: unbounded, like one near self-oscillation.
            if (L.motion & (kJumpAudio | kJumpNoise)) L.stable = false;
            L.overdriven = in_mag(m, SRACK_VCF_IN_AUDIO) > kLadderDriveMax;
            if (L.overdriven && hi > kLadderTameCutoff) L.stable = false;
            L.own = (L.overdriven || (L.motion & kJumpRare)) ? kLadderRareJumps : 1.0;
            if (in_motion(m, SRACK_VCF_IN_AUDIO) & kJumpNoise) L.own = res > kLadderNoiseInputRes ? kInf : L.own * kLadderNoiseInput;
            constexpr int kGrid = 12;
            for (int j = 0; j <= kGrid && L.stable; j++) {
                const double fr = lo + (hi - lo) * (double)j / kGrid;
                double l1[3];
                if (!ladder_l1(fr, res, l1) || l1[0] > kLadderL1Max) L.stable = false;
                for (int p = 0; p < 3; p++) L.l1[p] = std::max(L.l1[p], l1[p]);
                if (hi == lo) break;
            }
            if (!L.stable) L.l1[0] = L.l1[1] = L.l1[2] = kInf;
            const double audio = std::max(1.0, in_mag(m, SRACK_VCF_IN_AUDIO));
            for (int p = 0; p < 3; p++)
                L.cutoff[p] = !has_cv ? 0.0 : !L.stable ? kInf : 3.0 / std::max(lo, 0.01) * L.l1[p] * amount.abs_max() * audio;  // tools/ladder_calib.c l1: at most 2.7 / cutoff x L1
        }
    }

    // ---- 3. gains ---------------------------------------------------------------------------------------------------------------------
    bool is_event_input(int k, int i) const
    {
        switch (type(k)) {
        case SRACK_MOD_ADSR: return true;
        case SRACK_MOD_OSCILLATOR: return i == SRACK_OSC_IN_SYNC;
        case SRACK_MOD_GRID_SEQUENCER:
        case SRACK_MOD_PATTERN_SEQUENCER: return true;
        case SRACK_MOD_SAMPLE: return i == SRACK_SAMPLE_IN_GATE;
        default: return false;
        }
    }

    // d(output o of module k) / d(its input i): sup over the render, first order (a table, filled once the magnitudes and ladders are known)
    std::vector<std::vector<double>> edge_table;
    void edges()
    {
        edge_table.assign((size_t)n_mod, {});
        for (int k = 0; k < n_mod; k++) {
            if (!live[(size_t)k] || type(k) == SRACK_MOD_OUTPUT) continue;
            const Module& mod = g.modules[(size_t)k];
            const int n_out = (int)mag[(size_t)k].size();
            edge_table[(size_t)k].assign((size_t)(mod.n_in * n_out), 0.0);
            for (int i = 0; i < mod.n_in; i++)
                for (int o = 0; o < n_out; o++)
                    if (port_is_live(k, o)) edge_table[(size_t)k][(size_t)(i * n_out + o)] = edge_of(k, i, o);
        }
    }
    double edge(int k, int i, int o) const { return edge_table[(size_t)k][(size_t)i * mag[(size_t)k].size() + (size_t)o]; }
    double edge_of(int k, int i, int o) const
    {
        const Module& mod = g.modules[(size_t)k];
        const double out = std::max(1.0, mag[(size_t)k][(size_t)o]);
        if (is_event_input(k, i)) {
            const InputRef& in = mod.in[(size_t)i];
            if (in.src >= 0 && same_cycle(in.src, k)) return kInf;  // the moved event comes back to what produced it: the renders part for good
            return kEventGain * out;
        }
        switch (mod.type) {
        case SRACK_MOD_OSCILLATOR: {  // the pitch: delta = 440 * 2^(cv + val) / sr is summed into the phase, sample after sample
            const double delta = osc_delta_max(k);
            if (!(delta < kBig)) return kInf;
            const double phase = kApproxHorizon * 0.6931471805599453 * delta;  // cycles of phase per unit of CV after the horizon
            if (o == SRACK_OSC_OUT_SINE) return phase * 6.283185307179586;
            const bool aa = mod.fields[SRACK_OSC_ANTIALIASING] != 0.0;
            return aa ? phase / std::min(delta, 0.5) : phase * kEventGain;  // PolyBLEP's edge rises over 2 delta of phase; a raw edge is an event
        }
        case SRACK_MOD_MOOG_FILTER: return i == SRACK_VCF_IN_AUDIO ? ladder[(size_t)k].l1[o] : ladder[(size_t)k].cutoff[o];
        case SRACK_MOD_VCA: return connected(k, 0) && connected(k, 1) ? in_mag(k, 1 - i) : 0.0;  // vca.rs:132: audio * cv, continuous at cv = 0
        case SRACK_MOD_MONO_MIXER: return field(k, SRACK_MIX_GAIN0 + i).abs_max();
        case SRACK_MOD_MATH:
            if ((int)mod.fields[SRACK_MATH_OPERATION] != SRACK_MATH_MULTIPLY) return 1.0;
            if (i == 0) return connected(k, 1) ? in_mag(k, 1) : field(k, SRACK_MATH_CONSTANT).abs_max();
            return connected(k, 0) ? in_mag(k, 0) : 0.0;
        case SRACK_MOD_NONLINEAR: {  // sign(a) |a|^b, math.rs:203-205
            const double a = connected(k, 0) ? in_mag(k, 0) : 0.0;
            const Range b = nonlin_exponent(k);
            if (b.lo < 0.0 || !(a < kBig)) return kInf;
            if (i == 0) return b.lo < 1.0 ? kNonlinSteep : b.hi * std::pow(std::max(a, 1.0), b.hi - 1.0);
            return std::max(1.0, std::pow(std::max(a, 1.0), b.hi) * std::log(std::max(a, 2.718281828459045)));
        }
        case SRACK_MOD_SAMPLE: return kApproxHorizon * kEventGain * out;  // the pitch of a player without interpolation: a slipped index is an event
        case SRACK_MOD_FREEVERB: {
            if (mod.fields[SRACK_FREEVERB_FREEZE] != 0.0) return kInf;
            const double fb = std::min(std::fabs(mod.fields[SRACK_FREEVERB_ROOM_SIZE]) * 0.28 + 0.7, 0.999);
            return std::fabs(mod.fields[SRACK_FREEVERB_DRY]) + 3.0 * std::fabs(mod.fields[SRACK_FREEVERB_WET]) * 8.0 / (1.0 - fb) * 40.0;  // 8 combs, 4 allpasses (2.5 each)
        }
        default: return 0.0;
        }
    }

    using Gains = std::vector<std::vector<double>>;

    // the fixpoint for output channel c
    Gains gains_for(int c) const
    {
        Gains G((size_t)n_mod);
        for (int m = 0; m < n_mod; m++) G[(size_t)m].assign(mag[(size_t)m].size(), 0.0);
        std::vector<int> order;
        for (auto it = g.plan.order.rbegin(); it != g.plan.order.rend(); ++it)
            if (live[(size_t)*it] && *it != output) order.push_back(*it);
        auto wire = [&](int m, int p) {
            double s = 0.0;
            for (const Reader& r : readers[(size_t)m][(size_t)p]) {
                if (r.k == output) {
                    if (r.i == c) s += 1.0;
                    continue;
                }
                if (type(r.k) == SRACK_MOD_OUTPUT) continue;  // a second OutputModule is never heard
                for (int o = 0; o < (int)G[(size_t)r.k].size(); o++) {
                    if (!port_is_live(r.k, o)) continue;
                    const double go = G[(size_t)r.k][(size_t)o];
                    if (go == 0.0) continue;
                    const double e = edge(r.k, r.i, o);
                    if (e == 0.0) continue;
                    s += e * go;
                }
            }
            return s < kBig ? s : kInf;
        };
        Gains inc = G, inc_prev = G;  // the last two sweeps' increments per wire (zeros)
        bool changed = false;
        for (int sweep = 0; sweep < kSweeps + 8; sweep++) {
            changed = false;
            for (int m : order)
                for (int p = 0; p < (int)G[(size_t)m].size(); p++) {
                    if (!port_is_live(m, p)) continue;
                    double v = wire(m, p);
                    double& cur = G[(size_t)m][(size_t)p];
                    if (v > cur * (1.0 + 1e-3) && sweep >= kSweeps) v = kInf;  // still growing after kSweeps: a cycle with a loop gain of (about) one or more
                    if (v > cur * (1.0 + 1e-7)) changed = true;
                    inc_prev[(size_t)m][(size_t)p] = inc[(size_t)m][(size_t)p];
                    inc[(size_t)m][(size_t)p] = v > cur ? v - cur : 0.0;
                    if (v > cur) cur = v;
                }
            if (!changed) break;
        }
        // Left with something still creeping up (a loop gain around 0.99: 1e-7 ... 1e-3 per sweep): the rest of the geometric series its last two
        // increments imply is added, not dropped (the value was a few percent short, silently); increments that do not shrink: unbounded.
        if (changed)
            for (int m : order)
                for (int p = 0; p < (int)G[(size_t)m].size(); p++) {
                    double& cur = G[(size_t)m][(size_t)p];
                    const double d = inc[(size_t)m][(size_t)p], d0 = inc_prev[(size_t)m][(size_t)p];
                    if (d == 0.0 || cur == kInf) continue;
                    const double rho = d0 > 0.0 ? d / d0 : 1.0;
                    cur = rho < 1.0 ? cur + d * rho / (1.0 - rho) : kInf;
                    if (!(cur < kBig)) cur = kInf;
                }
        return G;
    }

    // (A feedback-FM loop — an oscillator whose own sine, through gains, comes back to its pitch: config 4 — is NOT cut out of the fixpoint.
    // In real arithmetic it is neutral: a phase perturbation psi obeys psi' = psi (1 + a cos(...)), a = delta ln2 2 pi |d cv / d sine| = 0.02
    // for config 4, whose logarithm averages to -a^2 / 4.  But the sine that travels round the loop is rounded to f32, and a phase
    // difference of 1e-12 flips one of those roundings now and then; each flip is a kick of 6e-8 on the pitch — far larger than what caused
    // it — and the kicks feed the difference that causes them.  Measured (tools/fm_sensitivity.c, the reference's own arithmetic twice,
    // one copy's phase off by 1e-12): nothing for 20 s, 2.6e-9 cycles at 25 s, 1e-7 at 35 s; and on the GPU (profiles/r05_horizon.json,
    // round 4's default kernels): config 4 at 4.6e-7 after one second, 1.5e-5 — outside the contract — after a minute, linear in between.
    // Only identical bits follow the reference round such a loop: first order says "diverges", and first order is right.)

    // The square of an oscillator that arrives UNCHANGED at an event input — wired straight to it, or handed on by a sequencer's gate outputs
    // (sequencer.rs:190-246: the gate output IS the step input where the cell is on) — costs nothing there: the default evaluation re-derives
    // any value close to zero with the reference's own operations (modules.hip.h, square_sign_safe), so its edges are the reference's.  The gain
    // of such a wire: event inputs it reaches unchanged count 0, everything else as usual.
    double pure_square_gain(const Gains& G, int m, int p, int c, int depth) const
    {
        double s = 0.0;
        for (const Reader& r : readers[(size_t)m][(size_t)p]) {
            if (r.k == output) {
                if (r.i == c) s += 1.0;
                continue;
            }
            if (type(r.k) == SRACK_MOD_OUTPUT) continue;
            const bool seq = type(r.k) == SRACK_MOD_GRID_SEQUENCER || type(r.k) == SRACK_MOD_PATTERN_SEQUENCER;
            if (is_event_input(r.k, r.i)) {
                if (same_cycle(m, r.k)) return kInf;
                if (seq && r.i == SRACK_SEQ_IN_STEP) {
                    const int first = type(r.k) == SRACK_MOD_GRID_SEQUENCER ? (int)SRACK_GRIDSEQ_OUT_GATE : (int)SRACK_PATSEQ_OUT_GATE0;
                    const int last = type(r.k) == SRACK_MOD_GRID_SEQUENCER ? (int)SRACK_GRIDSEQ_OUT_GATE : (int)SRACK_PATSEQ_OUT_GATE0 + 7;
                    for (int o = first; o <= last; o++) {
                        if (!port_is_live(r.k, o)) continue;
                        // (past four sequencers in a row the hand-on is no longer followed: that gate output counts with its ordinary wire gain —
                        // its event readers at kEventGain included —, not with 0 as until round 5: ADVICE r05)
                        s += depth < 4 ? pure_square_gain(G, r.k, o, c, depth + 1) : G[(size_t)r.k][(size_t)o];
                    }
                }
                continue;
            }
            for (int o = 0; o < (int)G[(size_t)r.k].size(); o++)
                if (port_is_live(r.k, o) && G[(size_t)r.k][(size_t)o] != 0.0) {
                    const double e = edge(r.k, r.i, o);
                    if (e != 0.0) s += e * G[(size_t)r.k][(size_t)o];
                }
        }
        return s;
    }
};

}  // namespace

int audible(const Graph& g, std::vector<char>& live, std::vector<uint32_t>& port_live, int* self_loop)
{
    const int n_mod = (int)g.modules.size();
    live.assign((size_t)n_mod, 0);
    port_live.assign((size_t)n_mod, 0);
    if (g.plan.output < 0) return 0;
    std::vector<int> stack{g.plan.output};
    while (!stack.empty()) {
        const int m = stack.back();
        stack.pop_back();
        if (live[(size_t)m]) continue;
        live[(size_t)m] = 1;
        for (const InputRef& in : g.modules[(size_t)m].in)
            if (in.src >= 0) {
                if (in.src == m) {
                    if (self_loop) *self_loop = m;
                    return -1;
                }
                port_live[(size_t)in.src] |= 1u << in.port;
                stack.push_back(in.src);
            }
    }
    return 0;
}

bool wire_sweeps(const Graph& g, int module)
{
    // iterative, each module once: 0 unseen, 1 on the current path, 2 done (holds), 3 done (sweeps)
    if (module < 0) return false;
    std::vector<char> state(g.modules.size(), 0);
    struct Frame { int m; size_t next; };
    std::vector<Frame> call{{module, 0}};
    auto leaf = [&](int m) -> int {  // 2 / 3 for a module that decides by itself, 0 for one that hands its inputs on
        switch (g.modules[(size_t)m].type) {
        case SRACK_MOD_ADSR:
        case SRACK_MOD_GRID_SEQUENCER:
        case SRACK_MOD_PATTERN_SEQUENCER: return 2;
        case SRACK_MOD_MATH:
        case SRACK_MOD_MONO_MIXER:
        case SRACK_MOD_VCA:
        case SRACK_MOD_NONLINEAR: return 0;
        default: return 3;  // oscillator, filter, noise, sample player, reverb
        }
    };
    if (int l = leaf(module)) return l == 3;
    state[(size_t)module] = 1;
    while (!call.empty()) {
        Frame& f = call.back();
        const Module& m = g.modules[(size_t)f.m];
        if (f.next < m.in.size()) {
            const int src = m.in[f.next++].src;
            if (src < 0) continue;
            if (state[(size_t)src] == 3 || state[(size_t)src] == 1) return true;  // a source that sweeps, or a feedback cycle of arithmetic
            if (state[(size_t)src] == 2) continue;
            const int l = leaf(src);
            if (l == 3) return true;
            if (l == 2) {
                state[(size_t)src] = 2;
                continue;
            }
            state[(size_t)src] = 1;
            call.push_back({src, 0});
        } else {
            state[(size_t)f.m] = 2;
            call.pop_back();
        }
    }
    return false;
}

ApproxPlan plan_approximations(const Graph& g, const std::vector<char>& live, const std::vector<uint32_t>& port_live,
                               const std::vector<VoiceOverride>& overrides, bool exact_requested)
{
    const int n_mod = (int)g.modules.size();
    ApproxPlan P;
    P.osc_exact.assign((size_t)n_mod, 0);
    P.exact_blep.assign((size_t)n_mod, 0);
    P.literal.assign((size_t)n_mod, 0);
    P.sine_loose.assign((size_t)n_mod, 0);
    P.nonlin_loose.assign((size_t)n_mod, 0);
    P.saw_fixed.assign((size_t)n_mod, 0);
    P.exact_patch = exact_requested;
    if (exact_requested || g.plan.output < 0) return P;

    Analysis A(g, live, port_live, overrides);
    A.components();
    A.magnitudes();
    A.ladders();
    A.edges();
    const int n_ch = std::min((int)g.modules[(size_t)g.plan.output].n_in, 8);
    std::vector<Analysis::Gains> G((size_t)n_ch);
    for (int c = 0; c < n_ch; c++)
        if (g.modules[(size_t)g.plan.output].in[(size_t)c].src >= 0) G[(size_t)c] = A.gains_for(c);
    P.mag = A.mag;
    P.gain.assign((size_t)n_mod, {});
    for (int m = 0; m < n_mod; m++) {
        P.gain[(size_t)m].assign(A.mag[(size_t)m].size(), 0.0);
        for (int c = 0; c < n_ch; c++)
            if (!G[(size_t)c].empty())
                for (size_t p = 0; p < P.gain[(size_t)m].size(); p++) P.gain[(size_t)m][p] = std::max(P.gain[(size_t)m][p], G[(size_t)c][(size_t)m][p]);
    }

    // ---- values without a bound: where they overflow, only the reference's own operations reproduce its infinities and NaNs (the default
    // forms clamp with v_med3, which sends a NaN to -1 where min / max send it to +1; round 4's seed 4386: two mixers feeding each other
    // with gains above one) ------------------------------------------------------------------------------------------------------------
    for (int m = 0; m < n_mod && !P.exact_patch; m++) {
        if (!live[(size_t)m] || m == g.plan.output) continue;
        for (size_t p = 0; p < A.mag[(size_t)m].size(); p++)
            if (A.port_is_live(m, (int)p) && A.mag[(size_t)m][p] == kInf) {
                P.exact_patch = true;
                P.unbounded_values = true;
                P.why = "unbounded values at module " + std::to_string(m) + " port " + std::to_string(p);
                break;
            }
    }
    // ---- an unbounded gain behind a module whose default evaluation is not the reference's to the bit -------------------------------------
    // An oscillator whose pitch moves (2^cv by polynomial) or whose sine is heard there (the polynomial sine: the reference's own but for 3
    // roundings in a million) is evaluated exactly as a whole — that oscillator: the libm's pow, the reference's sine, f64 PolyBLEP
    // (config 4: the modulator inside its feedback loop; the carrier behind it keeps the default forms).  The sample player's pitch has no
    // such form: the whole patch goes exact.  (A NonLinear does: denied its f32 form, its power is the host libm's own — modules.hip.h, powf_libm_plain.)
    for (int m = 0; m < n_mod; m++) {
        if (!live[(size_t)m]) continue;
        const Module& mod = g.modules[(size_t)m];
        const bool osc = mod.type == SRACK_MOD_OSCILLATOR && (mod.in[SRACK_OSC_IN_CV].src >= 0 || A.port_is_live(m, SRACK_OSC_OUT_SINE));
        const bool player = mod.type == SRACK_MOD_SAMPLE && mod.in[SRACK_SAMPLE_IN_CV].src >= 0;
        if (!osc && !player) continue;
        for (size_t p = 0; p < P.gain[(size_t)m].size(); p++)
            if (A.port_is_live(m, (int)p) && P.gain[(size_t)m][p] == kInf) {
                if (osc) {
                    P.osc_exact[(size_t)m] = 1;
                } else if (!P.exact_patch) {
                    P.exact_patch = true;
                    P.why = "unbounded gain behind module " + std::to_string(m) + " port " + std::to_string(p);
                }
            }
    }
    // (the decisions below are still made: SRACK_RENDER_KEEP_DEFAULT renders such a patch in the default flavour, module by module)

    // ---- the forms on offer, each with its contribution per channel ------------------------------------------------------------------
    enum Kind { kBlep, kSine, kFixed, kLadder, kNonlin };
    struct Form {
        int kind, module;
        std::vector<double> at;  // contribution per channel
        bool taken = true;
    };
    std::vector<Form> forms;
    auto add = [&](int kind, int module, auto&& per_channel) {
        Form f{kind, module, std::vector<double>((size_t)n_ch, 0.0), true};
        for (int c = 0; c < n_ch; c++)
            if (!G[(size_t)c].empty()) {
                const double v = per_channel(c);
                f.at[(size_t)c] = v < kBig ? v : kInf;
            }
        forms.push_back(std::move(f));
    };
    auto gw = [&](int c, int m, int p) { return G[(size_t)c][(size_t)m][(size_t)p]; };
    auto times = [](double eps, double gain) { return gain == 0.0 ? 0.0 : eps * gain; };
    for (int m = 0; m < n_mod; m++) {
        if (!live[(size_t)m]) continue;
        const Module& mod = g.modules[(size_t)m];
        if (mod.type == SRACK_MOD_OSCILLATOR) {
            const bool aa = mod.fields[SRACK_OSC_ANTIALIASING] != 0.0;
            const bool saw = A.port_is_live(m, SRACK_OSC_OUT_SAW), square = A.port_is_live(m, SRACK_OSC_OUT_SQUARE), sine = A.port_is_live(m, SRACK_OSC_OUT_SINE);
            if (aa && (saw || square))
                add(kBlep, m, [&](int c) {
                    double v = saw ? times(kEpsBlep, gw(c, m, SRACK_OSC_OUT_SAW)) : 0.0;
                    if (square) v += times(kEpsBlep, A.pure_square_gain(G[(size_t)c], m, SRACK_OSC_OUT_SQUARE, c, 0));
                    return v;
                });
            if (sine) add(kSine, m, [&](int c) { return times(kEpsSine, gw(c, m, SRACK_OSC_OUT_SINE)); });
            if (saw && !square && !sine && mod.in[SRACK_OSC_IN_CV].src < 0) {
                const double dt_min = 440.0 * std::exp2(A.field(m, SRACK_OSC_VAL).lo) / A.sr;  // (no CV: the pitch is val alone, per voice)
                const double eps = dt_min > 0.0 ? kEpsFixed + kEpsFixedWindow / dt_min : kInf;
                add(kFixed, m, [&](int c) { return times(eps, gw(c, m, SRACK_OSC_OUT_SAW)); });
            }
        } else if (mod.type == SRACK_MOD_MOOG_FILTER) {
            const Ladder& L = A.ladder[(size_t)m];
            const double j = !L.stable || L.overdriven || (L.motion & kJumpAudio) ? kInf : L.own;
            add(kLadder, m, [&](int c) {
                double v = 0.0;
                for (int p = 0; p < 3; p++)
                    if (A.port_is_live(m, p) && gw(c, m, p) != 0.0) v += j * kEpsLadder[p] * gw(c, m, p);
                return v;
            });
        } else if (mod.type == SRACK_MOD_NONLINEAR) {
            add(kNonlin, m, [&](int c) { return times(kEpsNonlin * std::max(1.0, A.mag[(size_t)m][0]), gw(c, m, 0)); });
        }
    }
    // A LITERAL ladder is only the reference's ladder bit for bit while its inputs are: behind an input that carries some taken form's error its
    // twenty f32 roundings per sample fall differently from the reference's, and the recurrence keeps those differences like the contracted
    // form's — the same epsilon, whatever the size of the input's error (tools/ladder_calib.c `gain`: a 2.4e-7 disturbance of the input comes
    // out at 0.8 / 2.3 / 0.9e-6 at resonance 0, 2.1 / 3.4 / 3.7e-6 at 0.8, where the static norms say 0.2 .. 2e-6; tools/cpu_soak.py found the
    // bound of seed 251484 — an f32 square into a literal bandpass — exceeded 2.6 x by exactly this).  Such a ladder's epsilon is a RESIDUAL:
    // it cannot be denied, only cleaned by denying what dirties its inputs.
    std::vector<double> residual((size_t)n_ch, 0.0);
    auto residuals = [&]() {
        // modules downstream of a taken form (the form's own module included)
        std::vector<char> dirty((size_t)n_mod, 0);
        std::vector<int> stack;
        for (const Form& f : forms)
            if (f.taken && !P.osc_exact[(size_t)f.module] && !dirty[(size_t)f.module]) dirty[(size_t)f.module] = 1, stack.push_back(f.module);
        while (!stack.empty()) {
            const int s_ = stack.back();
            stack.pop_back();
            for (int k = 0; k < n_mod; k++) {
                if (!live[(size_t)k] || dirty[(size_t)k]) continue;
                for (const InputRef& in : g.modules[(size_t)k].in)
                    if (in.src == s_) {
                        dirty[(size_t)k] = 1;
                        stack.push_back(k);
                        break;
                    }
            }
        }
        std::vector<double> r((size_t)n_ch, 0.0);
        for (const Form& f : forms) {
            if (f.kind != kLadder || f.taken) continue;
            const int m = f.module;
            const Ladder& L = A.ladder[(size_t)m];
            bool dirty_input = false;
            for (const InputRef& in : g.modules[(size_t)m].in)
                if (in.src >= 0 && live[(size_t)in.src] && dirty[(size_t)in.src]) dirty_input = true;
            if (!dirty_input || !L.stable) continue;  // (an unstable ladder's gains are unbounded: nothing in front of it is left to dirty it)
            const double j = L.own;  // (inf: a resonant ladder behind noise — nothing that perturbs its input may stay)
            for (int c = 0; c < n_ch; c++)
                if (!G[(size_t)c].empty())
                    for (int p_ = 0; p_ < 3; p_++)
                        if (A.port_is_live(m, p_) && gw(c, m, p_) != 0.0) {
                            const double v = j * kEpsLadder[p_] * gw(c, m, p_);
                            r[(size_t)c] += v < kBig ? v : kInf;
                        }
        }
        return r;
    };
    // Deny, largest first, until every channel is within the budget.  (A form that contributes nothing anywhere — nobody hears it — stays.)
    for (;;) {
        for (;;) {
            int worst_c = -1;
            double worst = kApproxBudget;
            for (int c = 0; c < n_ch; c++) {
                double s = residual[(size_t)c];
                bool deniable = false;
                for (const Form& f : forms)
                    if (f.taken) s += f.at[(size_t)c], deniable = deniable || f.at[(size_t)c] > 0.0;
                if (deniable && s > worst) worst = s, worst_c = c;  // (a channel over the budget by residuals alone has nothing left to deny)
            }
            if (worst_c < 0) break;
            Form* top = nullptr;
            for (Form& f : forms)
                if (f.taken && f.at[(size_t)worst_c] > 0.0 && (!top || f.at[(size_t)worst_c] > top->at[(size_t)worst_c])) top = &f;
            if (!top) break;
            top->taken = false;
            if (top->kind == kBlep)  // (the kernels' fixed-point saw IS the f32 PolyBLEP form: without that one there is no fixed-point phase to pay for)
                for (Form& f : forms)
                    if (f.kind == kFixed && f.module == top->module) f.taken = false;
        }
        // the residuals of what is taken NOW (they only shrink as forms are denied): consistent, or once more with them in the sums
        std::vector<double> now = residuals();
        bool fits = true;
        for (int c = 0; c < n_ch; c++) {
            double s = now[(size_t)c];
            for (const Form& f : forms)
                if (f.taken) s += f.at[(size_t)c];
            bool deniable = false;
            for (const Form& f : forms) deniable = deniable || (f.taken && f.at[(size_t)c] > 0.0);
            if (s > kApproxBudget && deniable) fits = false;
        }
        residual = now;
        if (fits) break;
    }
    for (const Form& f : forms) {
        switch (f.kind) {
        case kBlep: P.exact_blep[(size_t)f.module] = !f.taken; break;
        case kSine: P.sine_loose[(size_t)f.module] = f.taken; break;
        case kFixed: P.saw_fixed[(size_t)f.module] = f.taken; break;
        case kLadder: P.literal[(size_t)f.module] = !f.taken; break;
        case kNonlin: P.nonlin_loose[(size_t)f.module] = f.taken; break;
        }
    }
    // a fixed-point phase only makes sense with the f32 PolyBLEP (the kernels' fixed-point saw is that form); an exact oscillator has no other form
    for (int m = 0; m < n_mod; m++) {
        if (P.exact_blep[(size_t)m]) P.saw_fixed[(size_t)m] = 0;
        if (P.osc_exact[(size_t)m]) P.saw_fixed[(size_t)m] = P.sine_loose[(size_t)m] = 0;  // (exact_blep stays: what SRACK_RENDER_KEEP_DEFAULT falls back on)
    }
    for (int c = 0; c < n_ch; c++) {
        double s = residual[(size_t)c];
        for (const Form& f : forms)
            if (f.taken) s += f.at[(size_t)c];
        P.bound = std::max(P.bound, s);
    }
    return P;
}

}  // namespace srack
