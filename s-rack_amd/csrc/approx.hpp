// approx.hpp — which approximations of the default render mode a patch may take (flatten.cpp step 2b).
//
// The default mode's kernels have cheaper forms of four of the reference's computations — PolyBLEP in f32 (reference: f64,
// oscillator.rs:50-67), the ladder with one product of each a*b - c*d folded into an fma (filter.rs:69-82), the sine in f32, NonLinear's
// power through the f32 transcendental unit (math.rs:203-205) — and a constant-pitch saw's phase in 2^-64 fixed point.  Each is an error
// EPSILON injected on a wire; whether a patch may take it is decided here from a first-order bound: epsilon times the GAIN from that wire
// to every output channel, summed over the forms taken, must stay below kBudget (half the 1e-5 contract).  The gains come from one
// backward pass over the graph with a table per module type (approx.cpp): arithmetic passes errors on with its coefficients, a filter
// with the L1 norm of its impulse response (computed from its own coefficients: a ladder near self-oscillation has none), a pitch input
// INTEGRATES (gain ~ the render's length), an event input THRESHOLDS (gain ~ 1 / the probability that matters), a cycle multiplies by
// 1 / (1 - loop gain) or diverges.  Where a gain is unbounded every form in front of it is denied — an oscillator whose pitch moves, or whose
// sine is heard there, is then evaluated exactly as a whole (2^cv by the libm's pow, the reference's sine) —, and a patch whose VALUES have no
// bound, or with an unbounded gain behind a module without an exact form of its own, is rendered in the exact flavour altogether.
// Where no bound can be claimed the analysis says so instead of guessing: a ladder whose cutoff moves at audio rate, one driven above 1.75 with
// a cutoff that can pass 0.35 (the reference's ladder is chaotic there), a resonant one behind noise have unbounded gains — everything in front
// exact.  A literal ladder behind a perturbed input keeps the contracted one's epsilon as a residual that only cleaning its inputs removes.
// The constants (the forms' epsilons by input class and resonance, the drive limit, how envelopes, LFOs and sequencers move a cutoff) are
// MEASURED on a CPU emulation of both ladders (tools/ladder_calib.c; tests/test_approx.py re-runs it), and the whole bound is soaked on the CPU:
// tools/cpu_soak.py holds every fuzz patch, rendered by the oracle with the chosen forms emulated inside its modules, to its own bound.
#pragma once
#include <string>
#include <vector>

#include "flatten.hpp"

namespace srack {

constexpr double kApproxBudget = 5e-6;     // sup of the first-order error bound allowed at any output channel, in the contract's units (|gpu - ref| / max(|ref|, 1))
constexpr double kApproxHorizon = 2.88e7;  // samples the bound is derived for: ten minutes at 48 kHz (what integrates grows with the render's length)

struct ApproxPlan {
    // decisions, per module
    std::vector<char> osc_exact;     // oscillator: all of it as the reference spells it — 2^cv by the libm's pow, its sine, f64 PolyBLEP (OSC_EXACT on that op)
    std::vector<char> exact_blep;    // oscillator: f64 PolyBLEP (OSC_EXACT_BLEP)
    std::vector<char> literal;       // filter: the literal ladder (VCF_LITERAL)
    std::vector<char> sine_loose;    // oscillator: f32 sine (OSC_SINE_LOOSE)
    std::vector<char> nonlin_loose;  // NonLinear: f32 power (NONLIN_LOOSE)
    std::vector<char> saw_fixed;     // oscillator: phase in 2^-64 fixed point where the pitch is constant (OSC_FIXED_PHASE)
    bool exact_patch = false;        // the whole patch in the exact flavour (SRACK_RENDER_EXACT_OSC)
    std::string why;                 // what decided exact_patch ("" otherwise), for srack_render_info
    bool unbounded_values = false;   // ... it was values without a bound (overflow to inf / NaN): not waived by SRACK_RENDER_KEEP_DEFAULT
    // the analysis itself (tests, diagnostics)
    std::vector<std::vector<double>> mag;    // [module][output port]: sup |value| on the wire (inf: unbounded)
    std::vector<std::vector<double>> gain;   // [module][output port]: max over output channels of d(channel) / d(this wire) (inf: unbounded)
    double bound = 0.0;                      // the first-order bound at the worst channel with the decisions above
};

// flatten.cpp steps 1 and 2: the modules the plan's OutputModule can hear and which of their output ports anything reads.  Returns -1 and
// the module's index in `self_loop` for a module wired to itself (the reference deadlocks on it, synth.rs:99,251), else 0.
int audible(const Graph& g, std::vector<char>& live, std::vector<uint32_t>& port_live, int* self_loop);

// live / port_live: flatten.cpp steps 1 and 2 (which modules the output can hear, which of their ports anything reads).
ApproxPlan plan_approximations(const Graph& g, const std::vector<char>& live, const std::vector<uint32_t>& port_live,
                               const std::vector<VoiceOverride>& overrides, bool exact_requested);

// Does the wire SWEEP — an oscillator, a filter, noise, a sample player, a reverb somewhere upstream: a new value every sample — as opposed
// to HOLD (a sequencer's notes, an envelope, constants, arithmetic on those)?  (OSC_CV_AUDIO_RATE, program.hpp)
bool wire_sweeps(const Graph& g, int module);

}  // namespace srack
