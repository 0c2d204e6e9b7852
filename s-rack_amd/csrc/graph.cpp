// graph.cpp — module list, wiring and the planner.  See graph.hpp for the reference map.
#include "graph.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>

namespace srack {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* last_error() { return g_error.c_str(); }

// Module ids only have to be unique strings (they key the connection list of a .srk file, ui.rs:612-637); the layout
// follows a version-4 UUID so files written here look like the reference's.
uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

uint64_t noise_base_key(uint64_t seed, int module) { return splitmix64(seed ^ splitmix64((uint64_t)module)); }

std::string new_module_id()
{
    static std::atomic<uint64_t> counter{0};
    static const uint64_t seed = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
    auto mix = [](uint64_t x) {
        x += 0x9E3779B97F4A7C15ull;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    };
    const uint64_t n = counter.fetch_add(1);
    uint64_t a = mix(seed ^ (2 * n)), b = mix(seed ^ (2 * n + 1));
    a = (a & ~0xF000ull) | 0x4000ull;                        // version 4
    b = (b & ~(0xC0ull << 56)) | (0x80ull << 56);            // RFC 4122 variant
    char buf[40];
    std::snprintf(buf, sizeof buf, "%08x-%04x-%04x-%04x-%012llx", (unsigned)(a >> 32), (unsigned)((a >> 16) & 0xffff), (unsigned)(a & 0xffff),
                  (unsigned)(b >> 48), (unsigned long long)(b & 0xffffffffffffull));
    return buf;
}

int Graph::fields_of_type(int type)
{
    switch (type) {
    case SRACK_MOD_OUTPUT: return 0;
    case SRACK_MOD_OSCILLATOR: return SRACK_OSC__NFIELDS;
    case SRACK_MOD_MOOG_FILTER: return SRACK_VCF__NFIELDS;
    case SRACK_MOD_ADSR: return SRACK_ADSR__NFIELDS;
    case SRACK_MOD_VCA: return SRACK_VCA__NFIELDS;
    case SRACK_MOD_MONO_MIXER: return SRACK_MIX__NFIELDS;
    case SRACK_MOD_MATH: return SRACK_MATH__NFIELDS;
    case SRACK_MOD_GRID_SEQUENCER: return SRACK_GRIDSEQ__NFIELDS;
    case SRACK_MOD_PATTERN_SEQUENCER: return SRACK_PATSEQ__NFIELDS;
    case SRACK_MOD_NONLINEAR: return SRACK_NONLIN__NFIELDS;
    case SRACK_MOD_SAMPLE: return SRACK_SAMPLE__NFIELDS;
    case SRACK_MOD_NOISE: return 0;
    case SRACK_MOD_FREEVERB: return SRACK_FREEVERB__NFIELDS;
    default: return -1;
    }
}

bool Graph::field_is_state(int type, int field)
{
    switch (type) {
    case SRACK_MOD_OSCILLATOR: return field == SRACK_OSC_POS || field == SRACK_OSC_SYNC_LAST;
    case SRACK_MOD_MOOG_FILTER: return field >= SRACK_VCF_ST_F;
    case SRACK_MOD_ADSR:
        return field == SRACK_ADSR_PHASE || field == SRACK_ADSR_MODE || field == SRACK_ADSR_R_VAL || field == SRACK_ADSR_FROM_A_VAL ||
               field == SRACK_ADSR_GATE_LAST;
    case SRACK_MOD_GRID_SEQUENCER: return field >= SRACK_GRIDSEQ_CURRENT_STEP;
    case SRACK_MOD_PATTERN_SEQUENCER: return field >= SRACK_PATSEQ_CURRENT_STEP;
    case SRACK_MOD_SAMPLE: return field >= SRACK_SAMPLE_POS;
    default: return false;
    }
}

bool Graph::field_is_f64(int type, int field)
{
    return (type == SRACK_MOD_OSCILLATOR && field == SRACK_OSC_POS) || (type == SRACK_MOD_FREEVERB && field != SRACK_FREEVERB_FREEZE);
}

bool Graph::field_is_flag(int type, int field)
{
    switch (type) {
    case SRACK_MOD_OSCILLATOR: return field == SRACK_OSC_ANTIALIASING || field == SRACK_OSC_SYNC_LAST;
    case SRACK_MOD_ADSR: return field == SRACK_ADSR_MODE || field == SRACK_ADSR_GATE_LAST;
    case SRACK_MOD_VCA: return field == SRACK_VCA_NEGATIVE;
    case SRACK_MOD_MATH: return field == SRACK_MATH_OPERATION;
    case SRACK_MOD_GRID_SEQUENCER: return field != SRACK_GRIDSEQ_LAST;  // integers and detector bits; `last` is an f32
    case SRACK_MOD_PATTERN_SEQUENCER: return true;
    case SRACK_MOD_SAMPLE: return field == SRACK_SAMPLE_WAVE_NEW || field == SRACK_SAMPLE_PLAYING || field == SRACK_SAMPLE_GATE_LAST;
    case SRACK_MOD_FREEVERB: return field == SRACK_FREEVERB_FREEZE;
    default: return false;
    }
}

// Module::new(&audio_config)
int Graph::add_module(int type)
{
    Module m;
    m.type = type;
    int nf = fields_of_type(type);
    if (nf < 0) {
        set_error("add_module: module type " + std::to_string(type) + " is outside the hot-path scope");
        return SRACK_ERR_UNSUPPORTED;
    }
    m.fields.assign((size_t)nf, 0.0);
    switch (type) {
    case SRACK_MOD_OUTPUT:  // output.rs:15-23: one input per channel, no outputs
        m.n_in = (int)cfg.channels;
        m.n_out = 0;
        break;
    case SRACK_MOD_OSCILLATOR:  // oscillator.rs:27-41
        m.n_in = 2;
        m.n_out = 3;
        m.fields[SRACK_OSC_VAL] = 0.0;
        m.fields[SRACK_OSC_ANTIALIASING] = 1.0;
        m.fields[SRACK_OSC_POS] = 0.0;
        m.fields[SRACK_OSC_SYNC_LAST] = 1.0;  // TransitionDetector::new, synth.rs:283
        break;
    case SRACK_MOD_MOOG_FILTER:  // filter.rs:28-41; InternalMoogFilterState::default() = zeros
        m.n_in = 2;
        m.n_out = 3;
        m.fields[SRACK_VCF_FREQ] = (double)0.2f;
        m.fields[SRACK_VCF_RES] = (double)0.5f;
        m.fields[SRACK_VCF_EXP_AMT] = (double)0.5f;
        break;
    case SRACK_MOD_ADSR:  // adsr.rs:36-53
        m.n_in = 1;
        m.n_out = 1;
        m.fields[SRACK_ADSR_A_SEC] = 0.0;
        m.fields[SRACK_ADSR_D_SEC] = 0.5;
        m.fields[SRACK_ADSR_S_VAL] = 0.25;
        m.fields[SRACK_ADSR_R_SEC] = 0.5;
        m.fields[SRACK_ADSR_MODE] = SRACK_ADSR_MODE_NONE;
        m.fields[SRACK_ADSR_SAMPLE_RATE] = (double)(float)cfg.sample_rate;
        m.fields[SRACK_ADSR_GATE_LAST] = 1.0;
        break;
    case SRACK_MOD_VCA:  // vca.rs:18-26
        m.n_in = 2;
        m.n_out = 1;
        break;
    case SRACK_MOD_MONO_MIXER:  // mixer.rs:16-23
        m.n_in = 4;
        m.n_out = 1;
        for (int k = 0; k < 4; k++) m.fields[SRACK_MIX_GAIN0 + k] = 1.0;
        break;
    case SRACK_MOD_MATH:  // math.rs:26-35
        m.n_in = 2;
        m.n_out = 1;
        m.fields[SRACK_MATH_OPERATION] = SRACK_MATH_ADD;
        break;
    case SRACK_MOD_GRID_SEQUENCER:  // sequencer.rs:33-50: sequence = vec![None; 64], octaves 2, steps_per_octave 12
        m.n_in = 2;
        m.n_out = 3;
        m.fields[SRACK_GRIDSEQ_STEPS_PER_OCTAVE] = 12;
        m.fields[SRACK_GRIDSEQ_OCTAVES] = 2;
        m.fields[SRACK_GRIDSEQ_LENGTH] = 64;
        m.fields[SRACK_GRIDSEQ_STEP_LAST] = 1.0;  // both TransitionDetectors start at `true`
        m.fields[SRACK_GRIDSEQ_SYNC_LAST] = 1.0;
        m.cells.assign(64, 0u);
        break;
    case SRACK_MOD_PATTERN_SEQUENCER:  // sequencer.rs:352-368: 8 gate outputs + sync, sequence = vec![vec![None; 64]; 8]
        m.n_in = 2;
        m.n_out = 9;
        m.fields[SRACK_PATSEQ_LENGTH] = 64;
        m.fields[SRACK_PATSEQ_STEP_LAST] = 1.0;
        m.fields[SRACK_PATSEQ_SYNC_LAST] = 1.0;
        m.cells.assign(64, 0u);
        break;
    case SRACK_MOD_NONLINEAR:  // math.rs:186-196
        m.n_in = 2;
        m.n_out = 1;
        m.fields[SRACK_NONLIN_CONSTANT] = 1.0;
        break;
    case SRACK_MOD_SAMPLE:  // sample.rs:88-101; WaveBox::default(): no samples, sample_rate 0.0, new = false
        m.n_in = 2;
        m.n_out = 1;
        m.fields[SRACK_SAMPLE_SAMPLE_RATE] = (double)(float)cfg.sample_rate;
        m.fields[SRACK_SAMPLE_GATE_LAST] = 1.0;
        break;
    case SRACK_MOD_NOISE:  // oscillator.rs:314-320: no inputs, one output, no parameters
        m.n_in = 0;
        m.n_out = 1;
        break;
    case SRACK_MOD_FREEVERB:  // freeverb.rs:60-82
        m.n_in = 2;
        m.n_out = 2;
        m.fields[SRACK_FREEVERB_DAMPENING] = 0.5;
        m.fields[SRACK_FREEVERB_WET] = 1.0;
        m.fields[SRACK_FREEVERB_WIDTH] = 0.5;
        m.fields[SRACK_FREEVERB_ROOM_SIZE] = 0.5;
        break;
    }
    m.in.assign((size_t)m.n_in, InputRef{});
    m.id = new_module_id();
    modules.push_back(std::move(m));
    plan.valid = false;
    revision++;
    return (int)modules.size() - 1;
}

int Graph::num_fields(int module) const
{
    if (module < 0 || module >= (int)modules.size()) return SRACK_ERR_INVALID;
    return (int)modules[(size_t)module].fields.size();
}

int Graph::set_field(int module, int field, double value)
{
    int nf = num_fields(module);
    if (nf < 0 || field < 0 || field >= nf) {
        set_error("set_field: no such module/field");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)module];
    if ((m.type == SRACK_MOD_GRID_SEQUENCER && field == SRACK_GRIDSEQ_LENGTH) || (m.type == SRACK_MOD_PATTERN_SEQUENCER && field == SRACK_PATSEQ_LENGTH)) {
        if (!(value >= 1.0 && value <= 64.0)) {  // the UI keeps 2..64 (sequencer.rs:101-131); an empty sequence would index out of bounds
            set_error("set_field: sequence length must be 1..64");
            return SRACK_ERR_INVALID;
        }
    }
    if (field_is_f64(m.type, field))
        m.fields[(size_t)field] = value;
    else if (field_is_flag(m.type, field))
        m.fields[(size_t)field] = !(value == value) ? 0.0 : (double)(int)std::fmin(std::fmax(value, -2147483648.0), 2147483647.0);  // (int) of a NaN / out-of-range double is undefined
    else
        m.fields[(size_t)field] = (double)(float)value;  // the struct member is an f32
    revision++;
    return SRACK_OK;
}

int Graph::get_field(int module, int field, double* value) const
{
    int nf = num_fields(module);
    if (nf < 0 || field < 0 || field >= nf || !value) {
        set_error("get_field: no such module/field");
        return SRACK_ERR_INVALID;
    }
    *value = modules[(size_t)module].fields[(size_t)field];
    return SRACK_OK;
}

// A grid cell of a sequencer (what the egui editors toggle, sequencer.rs:137-184, 437-478).
int Graph::set_step(int module, int channel, int step, int state, int value)
{
    if (module < 0 || module >= (int)modules.size()) {
        set_error("set_step: no such module");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)module];
    const bool grid = m.type == SRACK_MOD_GRID_SEQUENCER;
    if ((!grid && m.type != SRACK_MOD_PATTERN_SEQUENCER) || step < 0 || step >= 64 || channel < 0 || channel >= (grid ? 1 : 8) || state < 0 ||
        state > SRACK_STEP_HOLD || value < 0 || value > 65535) {
        set_error("set_step: not a sequencer, or channel / step / state / value out of range");
        return SRACK_ERR_INVALID;
    }
    uint32_t& cell = m.cells[(size_t)step];
    if (grid) {
        cell = state == SRACK_STEP_NONE ? 0u : (0x80000000u | (state == SRACK_STEP_HOLD ? 0x40000000u : 0u) | (uint32_t)value);
    } else {
        cell &= ~(3u << (2 * channel));
        if (state != SRACK_STEP_NONE) cell |= (1u | (state == SRACK_STEP_HOLD ? 2u : 0u)) << (2 * channel);
    }
    revision++;
    return SRACK_OK;
}

int Graph::get_step(int module, int channel, int step, int* state, int* value) const
{
    if (module < 0 || module >= (int)modules.size()) {
        set_error("get_step: no such module");
        return SRACK_ERR_INVALID;
    }
    const Module& m = modules[(size_t)module];
    const bool grid = m.type == SRACK_MOD_GRID_SEQUENCER;
    if ((!grid && m.type != SRACK_MOD_PATTERN_SEQUENCER) || step < 0 || step >= 64 || channel < 0 || channel >= (grid ? 1 : 8)) {
        set_error("get_step: not a sequencer, or channel / step out of range");
        return SRACK_ERR_INVALID;
    }
    const uint32_t cell = m.cells[(size_t)step];
    int st, v = 0;
    if (grid) {
        st = !(cell & 0x80000000u) ? SRACK_STEP_NONE : ((cell & 0x40000000u) ? SRACK_STEP_HOLD : SRACK_STEP_ON);
        v = (int)(cell & 0xffffu);
    } else {
        const uint32_t b = (cell >> (2 * channel)) & 3u;
        st = !(b & 1u) ? SRACK_STEP_NONE : ((b & 2u) ? SRACK_STEP_HOLD : SRACK_STEP_ON);
    }
    if (state) *state = st;
    if (value) *value = v;
    return SRACK_OK;
}

// WaveBox::load's result (sample.rs:31-69): samples, sample_rate, new = true
int Graph::set_wave(int module, const float* samples, uint32_t n, float sample_rate)
{
    if (module < 0 || module >= (int)modules.size() || modules[(size_t)module].type != SRACK_MOD_SAMPLE || (n && !samples)) {
        set_error("set_wave: not a SampleModule");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)module];
    m.wave.assign(samples, samples + n);
    m.fields[SRACK_SAMPLE_WAVE_SAMPLE_RATE] = (double)sample_rate;
    m.fields[SRACK_SAMPLE_WAVE_NEW] = 1.0;
    revision++;
    modules[(size_t)module].wave_revision = revision;
    return SRACK_OK;
}

// Contents of one output buffer as a loaded .srk leaves them: only observable through a broken feedback edge, whose
// sink reads the source's buffer before the source has run.
int Graph::set_output_buffer(int module, int port, const float* samples, uint32_t n)
{
    if (module < 0 || module >= (int)modules.size() || port < 0 || port >= modules[(size_t)module].n_out || (n != 0 && n != cfg.buffer_size) ||
        (n && !samples)) {
        set_error("set_output_buffer: no such module / port, or length != buffer_size");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)module];
    m.out_init.resize((size_t)m.n_out);
    m.out_init[(size_t)port].assign(samples, samples + n);
    revision++;
    return SRACK_OK;
}

// SynthModule::set_input
int Graph::connect(int src, int src_port, int sink, int sink_port)
{
    int n = (int)modules.size();
    if (src < 0 || src >= n || sink < 0 || sink >= n) {
        set_error("connect: module index out of range");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)sink];
    if (sink_port < 0 || sink_port >= m.n_in || src_port < 0 || src_port >= modules[(size_t)src].n_out) {
        set_error("connect: port index out of range");  // Err(()) of set_input / get_output
        return SRACK_ERR_PORT;
    }
    m.in[(size_t)sink_port] = InputRef{src, src_port};
    plan.valid = false;
    revision++;
    return SRACK_OK;
}

int Graph::disconnect(int sink, int sink_port)
{
    if (sink < 0 || sink >= (int)modules.size()) {
        set_error("disconnect: module index out of range");
        return SRACK_ERR_INVALID;
    }
    Module& m = modules[(size_t)sink];
    if (sink_port < 0 || sink_port >= m.n_in) {
        set_error("disconnect: port index out of range");
        return SRACK_ERR_PORT;
    }
    m.in[(size_t)sink_port] = InputRef{};
    plan.valid = false;
    revision++;
    return SRACK_OK;
}

// ---- plan_execution ---------------------------------------------------------------------------

using Adjacency = std::vector<std::vector<int>>;  // sink -> sources, input-index order, duplicates kept

// is_loop (synth.rs:107-126): breadth-first over `to_search` in insertion order; returns the first
// node found that still lists `module` as a source.
static int find_loop_closer(int module, const Adjacency& sources)
{
    std::vector<int> frontier{module};
    std::vector<char> seen(sources.size(), 0);
    size_t scan = 0;
    for (;;) {
        // first not-yet-visited entry of the list; entries before `scan` are known visited
        while (scan < frontier.size() && seen[(size_t)frontier[scan]]) scan++;
        if (scan == frontier.size()) return -1;
        int current = frontier[scan];
        seen[(size_t)current] = 1;
        for (int dep : sources[(size_t)current]) {
            if (dep == module) return current;
            frontier.push_back(dep);
        }
    }
}

int Graph::make_plan(int output, const std::vector<int>& all_modules)
{
    const size_t n = modules.size();
    plan = Plan{};
    plan.output = output;
    plan.position.assign(n, -1);
    // phase 1 (synth.rs:134-163): scheduler edges for every module reachable from the list
    Adjacency sources(n);
    {
        std::vector<char> seen(n, 0);
        std::vector<int> stack(all_modules);
        stack.push_back(output);
        while (!stack.empty()) {
            int m = stack.back();
            stack.pop_back();
            if (seen[(size_t)m]) continue;
            seen[(size_t)m] = 1;
            for (const InputRef& in : modules[(size_t)m].in)
                if (in.src >= 0) {
                    stack.push_back(in.src);
                    sources[(size_t)m].push_back(in.src);
                }
        }
    }
    // phase 2 (synth.rs:164-192): depth-first from the output; for every newly visited module
    // drop the edges that close a cycle through it
    {
        std::vector<char> seen(n, 0);
        std::vector<int> stack(all_modules);
        stack.push_back(output);
        while (!stack.empty()) {
            int m = stack.back();
            stack.pop_back();
            if (seen[(size_t)m]) continue;
            seen[(size_t)m] = 1;
            stack.insert(stack.end(), sources[(size_t)m].begin(), sources[(size_t)m].end());
            for (int from; (from = find_loop_closer(m, sources)) >= 0;) {
                auto& deps = sources[(size_t)from];
                deps.erase(std::remove(deps.begin(), deps.end(), m), deps.end());
                plan.removed.emplace_back(from, m);
            }
        }
    }
    // phase 3 (synth.rs:193-211): first unvisited list entry whose sources are all visited
    {
        std::vector<char> done(n, 0);
        for (;;) {
            int pick = -1;
            for (int m : all_modules) {
                if (done[(size_t)m]) continue;
                bool ready = true;
                for (int dep : sources[(size_t)m])
                    if (!done[(size_t)dep]) {
                        ready = false;
                        break;
                    }
                if (ready) {
                    pick = m;
                    break;
                }
            }
            if (pick < 0) break;
            done[(size_t)pick] = 1;
            plan.position[(size_t)pick] = (int)plan.order.size();
            plan.order.push_back(pick);
        }
    }
    plan.valid = true;
    return (int)plan.order.size();
}

int Graph::make_plan()
{
    int output = -1;  // find_output: first OutputModule in list order (ui.rs:84-96)
    for (size_t i = 0; i < modules.size(); i++)
        if (modules[i].type == SRACK_MOD_OUTPUT) {
            output = (int)i;
            break;
        }
    if (output < 0) {  // ui.rs:75-79: plan cleared, output = None
        plan = Plan{};
        plan.position.assign(modules.size(), -1);
        plan.valid = true;
        return 0;
    }
    std::vector<int> all(modules.size());
    for (size_t i = 0; i < all.size(); i++) all[i] = (int)i;
    return make_plan(output, all);
}

std::vector<Edge> Graph::delayed_edges() const
{
    std::vector<Edge> out;
    if (!plan.valid) return out;
    for (int sink : plan.order) {
        const Module& m = modules[(size_t)sink];
        for (int k = 0; k < m.n_in; k++) {
            const InputRef& in = m.in[(size_t)k];
            if (in.src < 0) continue;
            int ps = plan.position[(size_t)in.src];
            if (ps >= 0 && ps > plan.position[(size_t)sink]) out.push_back(Edge{in.src, in.port, sink, k});
        }
    }
    return out;
}

}  // namespace srack
