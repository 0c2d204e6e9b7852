// srk.cpp — the .srk rack file <-> Graph.
//
// A .srk is `FileFormat { modules, connections, positions }` (ui.rs:578-586) written with
// `rmp_serde::Serializer::new` (ui.rs:112): MessagePack in rmp-serde 1.3.0's default ("compact") form —
//   struct            -> array of its non-skipped fields in declaration order
//   newtype struct    -> its inner value (AudioBuffer -> nil | array of f32)
//   enum newtype variant -> map { "VariantName": value }     (SynthModuleType, synth.rs:300-317)
//   enum unit variant -> "VariantName"                        (ADSRMode adsr.rs:27-33, MathOperation math.rs:7-11)
//   Option            -> nil | value;  tuple / Vec / [T; N] / Box<[T]> -> array;  Arc / RwLock / Mutex -> inner value
//   integers in the shortest encoding, f32 as float32, f64 as float64.
// The per-module field lists below are the reference's struct definitions (cited per case).
//
// Loading follows SynthModuleWorkspaceImpl::deserialize (ui.rs:116-135): modules are popped off the END of the
// file's list, so the workspace order — which the planner depends on — is the file order REVERSED (ui.rs:654-660);
// V0 variants migrate (sequencer.rs:647-670, filter.rs:265-281); set_audio_config is applied to every module (the
// oscillator and the sample player take the host's sample rate, the ADSR keeps the saved one, adsr.rs:69-71; a saved
// buffer survives only if its length is the host's buffer_size); connections are applied from the end of the list,
// silently skipping unknown ids and bad ports (ui.rs:672-680).
//
// PARITY: the reference ships no .srk file and no round-trip test, and rmp-serde is not in the tree, so this format
// reading is pinned by nothing the reference holds ("parity unpinned"); tests/test_srk.py checks
// it against an independent MessagePack codec and against patches built through the graph API.
#include <cmath>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "graph.hpp"

namespace srack {

namespace {

struct ParseError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct UnsupportedError : ParseError {
    using ParseError::ParseError;
};

// ---- MessagePack reader ---------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t* p;
    const uint8_t* end;

    [[noreturn]] void fail(const std::string& what) const { throw ParseError("srk: " + what); }
    void need(size_t n) const
    {
        if ((size_t)(end - p) < n) fail("unexpected end of file");
    }
    uint8_t peek() const
    {
        need(1);
        return *p;
    }
    uint64_t be(int n)
    {
        need((size_t)n);
        uint64_t v = 0;
        for (int i = 0; i < n; i++) v = (v << 8) | *p++;
        return v;
    }
    bool nil()
    {
        if (peek() != 0xc0) return false;
        p++;
        return true;
    }
    bool boolean()
    {
        const uint8_t t = peek();
        if (t != 0xc2 && t != 0xc3) fail("expected a bool");
        p++;
        return t == 0xc3;
    }
    bool is_int() const
    {
        const uint8_t t = peek();
        return t <= 0x7f || t >= 0xe0 || (t >= 0xcc && t <= 0xd3);
    }
    int64_t integer()
    {
        const uint8_t t = peek();
        p++;
        if (t <= 0x7f) return t;
        if (t >= 0xe0) return (int8_t)t;
        switch (t) {
        case 0xcc: return (int64_t)be(1);
        case 0xcd: return (int64_t)be(2);
        case 0xce: return (int64_t)be(4);
        case 0xcf: return (int64_t)be(8);
        case 0xd0: return (int8_t)be(1);
        case 0xd1: return (int16_t)be(2);
        case 0xd2: return (int32_t)be(4);
        case 0xd3: return (int64_t)be(8);
        }
        fail("expected an integer");
    }
    int64_t integer_in(int64_t lo, int64_t hi, const char* what)
    {
        const int64_t v = integer();
        if (v < lo || v > hi) fail(std::string(what) + " out of range");
        return v;
    }
    double number()  // serde's numeric visitors accept any MessagePack number for an f32 / f64 field
    {
        const uint8_t t = peek();
        if (t == 0xca) {
            p++;
            const uint32_t u = (uint32_t)be(4);
            float f;
            std::memcpy(&f, &u, 4);
            return (double)f;
        }
        if (t == 0xcb) {
            p++;
            const uint64_t u = be(8);
            double d;
            std::memcpy(&d, &u, 8);
            return d;
        }
        if (is_int()) return (double)integer();
        fail("expected a number");
    }
    std::string str()
    {
        const uint8_t t = peek();
        p++;
        size_t n;
        if ((t & 0xe0) == 0xa0)
            n = t & 0x1f;
        else if (t == 0xd9)
            n = (size_t)be(1);
        else if (t == 0xda)
            n = (size_t)be(2);
        else if (t == 0xdb)
            n = (size_t)be(4);
        else
            fail("expected a string");
        need(n);
        std::string s((const char*)p, n);
        p += n;
        return s;
    }
    uint32_t array()
    {
        const uint8_t t = peek();
        p++;
        if ((t & 0xf0) == 0x90) return t & 0x0f;
        if (t == 0xdc) return (uint32_t)be(2);
        if (t == 0xdd) return (uint32_t)be(4);
        fail("expected an array (structs are written in rmp-serde's compact form)");
    }
    void array_of(uint32_t n, const char* what)
    {
        if (array() != n) fail(std::string(what) + ": wrong number of fields");
    }
    uint32_t map()
    {
        const uint8_t t = peek();
        p++;
        if ((t & 0xf0) == 0x80) return t & 0x0f;
        if (t == 0xde) return (uint32_t)be(2);
        if (t == 0xdf) return (uint32_t)be(4);
        fail("expected a map");
    }
    std::vector<float> f32_array()
    {
        const uint32_t n = array();
        need(n);  // at least one byte per element: bounds the allocation by the file size
        std::vector<float> v(n);
        for (uint32_t i = 0; i < n; i++) v[i] = (float)number();
        return v;
    }
    std::vector<float> audio_buffer()  // AudioBuffer(Option<Arc<RwLock<Box<[f32]>>>>), synth.rs:27-28; None => empty
    {
        if (nil()) return {};
        return f32_array();
    }
    bool transition_detector()  // struct TransitionDetector { last: bool }, synth.rs:276-279
    {
        array_of(1, "TransitionDetector");
        return boolean();
    }
};

// A saved buffer survives the load only when set_audio_config's resize() leaves it alone (synth.rs:35-44).
void keep_buffer(Module& m, int port, std::vector<float>&& buf, const AudioConfig& cfg)
{
    if (buf.size() != (size_t)cfg.buffer_size) return;
    bool any = false;
    for (float f : buf) any = any || f != 0.0f || std::signbit(f);
    if (!any) return;  // all +0.0: the same as a fresh buffer
    m.out_init.resize((size_t)m.n_out);
    m.out_init[(size_t)port] = std::move(buf);
}

int index_of(const char* const* names, int n, const std::string& s)
{
    for (int i = 0; i < n; i++)
        if (s == names[i]) return i;
    return -1;
}

const char* const kAdsrModes[] = {"Attack", "Decay", "Sustain", "Release", "None"};  // adsr.rs:27-33
const char* const kMathOps[] = {"Add", "Subtract", "Multiply"};                      // math.rs:7-11

Module parse_module(Reader& r, Graph& scratch)
{
    if (r.map() != 1) r.fail("SynthModuleType: expected { variant: module }");
    const std::string variant = r.str();
    const AudioConfig& cfg = scratch.cfg;
    auto fresh = [&](int type) {  // Module::new defaults for everything the file does not carry
        scratch.modules.clear();
        const int rc = scratch.add_module(type);
        if (rc < 0) r.fail(last_error());
        return scratch.modules.back();
    };
    auto f32 = [&]() { return (double)(float)r.number(); };

    if (variant == "OutputModuleV0") {  // output.rs:6-12: id, bufs
        r.array_of(2, "OutputModule");
        Module m = fresh(SRACK_MOD_OUTPUT);  // set_audio_config rebuilds inputs and bufs for the host's channels (output.rs:39-44)
        m.id = r.str();
        for (uint32_t i = 0, n = r.array(); i < n; i++) r.audio_buffer();
        return m;
    }
    if (variant == "OscillatorModuleV0") {  // oscillator.rs:9-24: id, val, sample_rate, sine, square, saw, pos, antialiasing, sync_detector
        r.array_of(9, "OscillatorModule");
        Module m = fresh(SRACK_MOD_OSCILLATOR);
        m.id = r.str();
        m.fields[SRACK_OSC_VAL] = f32();
        r.integer_in(0, 65535, "sample_rate");  // replaced by the host's rate (oscillator.rs:84)
        for (int port = 0; port < 3; port++) keep_buffer(m, port, r.audio_buffer(), cfg);
        m.fields[SRACK_OSC_POS] = r.number();
        m.fields[SRACK_OSC_ANTIALIASING] = r.boolean();
        m.fields[SRACK_OSC_SYNC_LAST] = r.transition_detector();
        return m;
    }
    if (variant == "MoogFilterModuleV1" || variant == "MoogFilterModuleV0") {
        // filter.rs:11-25: id, lowpass, bandpass, highpass, freq, res, exp_amt, state;  V0 (filter.rs:252-263): id, buf, ...
        const bool v0 = variant == "MoogFilterModuleV0";
        r.array_of(v0 ? 6 : 8, "MoogFilterModule");
        Module m = fresh(SRACK_MOD_MOOG_FILTER);
        m.id = r.str();
        if (v0) {
            if (r.peek() == 0xc0) r.fail("MoogFilterModuleV0 without a buffer (the reference unwraps it, filter.rs:267)");
            keep_buffer(m, 0, r.audio_buffer(), cfg);  // buf becomes lowpass; bandpass / highpass are new
        } else {
            for (int port = 0; port < 3; port++) keep_buffer(m, port, r.audio_buffer(), cfg);
        }
        m.fields[SRACK_VCF_FREQ] = f32();
        m.fields[SRACK_VCF_RES] = f32();
        m.fields[SRACK_VCF_EXP_AMT] = f32();
        r.array_of(6, "InternalMoogFilterState");  // filter.rs:48-56: f, p, q, b[5], freq, res
        m.fields[SRACK_VCF_ST_F] = f32();
        m.fields[SRACK_VCF_ST_P] = f32();
        m.fields[SRACK_VCF_ST_Q] = f32();
        r.array_of(5, "InternalMoogFilterState.b");
        for (int k = 0; k < 5; k++) m.fields[SRACK_VCF_ST_B0 + k] = f32();
        m.fields[SRACK_VCF_ST_FREQ] = f32();
        m.fields[SRACK_VCF_ST_RES] = f32();
        return m;
    }
    if (variant == "ADSRModuleV0") {
        // adsr.rs:7-24: id, a_sec, d_sec, s_val, r_sec, phase, mode, r_val, from_a_val, sample_rate, transition_detector, output_buffer, ui_dirty
        r.array_of(13, "ADSRModule");
        Module m = fresh(SRACK_MOD_ADSR);
        m.id = r.str();
        m.fields[SRACK_ADSR_A_SEC] = f32();
        m.fields[SRACK_ADSR_D_SEC] = f32();
        m.fields[SRACK_ADSR_S_VAL] = f32();
        m.fields[SRACK_ADSR_R_SEC] = f32();
        m.fields[SRACK_ADSR_PHASE] = f32();
        const int mode = index_of(kAdsrModes, 5, r.str());
        if (mode < 0) r.fail("ADSRMode: unknown variant");
        m.fields[SRACK_ADSR_MODE] = mode;
        m.fields[SRACK_ADSR_R_VAL] = f32();
        m.fields[SRACK_ADSR_FROM_A_VAL] = f32();
        m.fields[SRACK_ADSR_SAMPLE_RATE] = f32();  // kept: set_audio_config does not refresh it (adsr.rs:69-71)
        m.fields[SRACK_ADSR_GATE_LAST] = r.transition_detector();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        r.boolean();  // ui_dirty
        return m;
    }
    if (variant == "VCAModuleV0") {  // vca.rs:6-15: id, buf, negative
        r.array_of(3, "VCAModule");
        Module m = fresh(SRACK_MOD_VCA);
        m.id = r.str();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        m.fields[SRACK_VCA_NEGATIVE] = r.boolean();
        return m;
    }
    if (variant == "MonoMixerModuleV0") {  // mixer.rs:6-13: id, gain, buf; the number of inputs is gain.len() (mixer.rs:40)
        r.array_of(3, "MonoMixerModule");
        Module m = fresh(SRACK_MOD_MONO_MIXER);
        m.id = r.str();
        const std::vector<float> gain = r.f32_array();
        if (gain.size() > 4) r.fail("MonoMixerModule with more than 4 inputs");
        m.n_in = (int)gain.size();
        m.in.assign((size_t)m.n_in, InputRef{});
        for (size_t k = 0; k < gain.size(); k++) m.fields[SRACK_MIX_GAIN0 + k] = (double)gain[k];
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        return m;
    }
    if (variant == "MathModuleV0") {  // math.rs:13-23: id, buf, constant, operation
        r.array_of(4, "MathModule");
        Module m = fresh(SRACK_MOD_MATH);
        m.id = r.str();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        m.fields[SRACK_MATH_CONSTANT] = f32();
        const int op = index_of(kMathOps, 3, r.str());
        if (op < 0) r.fail("MathOperation: unknown variant");
        m.fields[SRACK_MATH_OPERATION] = op;
        return m;
    }
    if (variant == "NonLinearModuleV0") {  // math.rs:176-185: id, buf, constant
        r.array_of(3, "NonLinearModule");
        Module m = fresh(SRACK_MOD_NONLINEAR);
        m.id = r.str();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        m.fields[SRACK_NONLIN_CONSTANT] = f32();
        return m;
    }
    if (variant == "SampleModuleV0") {  // sample.rs:72-85: id, transition_detector, pos, buf, wavebox, playing, sample_rate
        r.array_of(7, "SampleModule");
        Module m = fresh(SRACK_MOD_SAMPLE);
        m.id = r.str();
        m.fields[SRACK_SAMPLE_GATE_LAST] = r.transition_detector();
        m.fields[SRACK_SAMPLE_POS] = f32();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        r.array_of(3, "WaveBox");  // sample.rs:15-20: samples, sample_rate, new
        m.wave = r.f32_array();
        m.fields[SRACK_SAMPLE_WAVE_SAMPLE_RATE] = f32();
        m.fields[SRACK_SAMPLE_WAVE_NEW] = r.boolean();
        m.fields[SRACK_SAMPLE_PLAYING] = r.boolean();
        r.number();  // sample_rate: replaced by the host's (sample.rs:117)
        return m;
    }
    if (variant == "GridSequencerModuleV1" || variant == "GridSequencerModuleV0") {
        // sequencer.rs:12-30: id, cv_out, gate_out, sync_out, sequence, octaves, steps_per_octave, current_step,
        // transition_detector, sync_transition_detector, last, ui_dirty;  V0: sequence of Option<u16> (sequencer.rs:625-645)
        const bool v0 = variant == "GridSequencerModuleV0";
        r.array_of(12, "GridSequencerModule");
        Module m = fresh(SRACK_MOD_GRID_SEQUENCER);
        m.id = r.str();
        for (int port = 0; port < 3; port++) keep_buffer(m, port, r.audio_buffer(), cfg);
        const uint32_t len = r.array();
        if (len < 1 || len > 64) r.fail("GridSequencerModule: sequence length must be 1..64");
        m.fields[SRACK_GRIDSEQ_LENGTH] = len;
        for (uint32_t i = 0; i < len; i++) {
            if (r.nil()) continue;
            uint32_t value;
            bool hold = false;
            if (v0) {
                value = (uint32_t)r.integer_in(0, 65535, "note");
            } else {
                r.array_of(2, "(u16, bool)");
                value = (uint32_t)r.integer_in(0, 65535, "note");
                hold = r.boolean();
            }
            m.cells[i] = 0x80000000u | (hold ? 0x40000000u : 0u) | value;
        }
        m.fields[SRACK_GRIDSEQ_OCTAVES] = (double)r.integer_in(0, 255, "octaves");
        m.fields[SRACK_GRIDSEQ_STEPS_PER_OCTAVE] = (double)r.integer_in(0, 65535, "steps_per_octave");
        m.fields[SRACK_GRIDSEQ_CURRENT_STEP] = (double)r.integer_in(0, 65535, "current_step");
        m.fields[SRACK_GRIDSEQ_STEP_LAST] = r.transition_detector();
        m.fields[SRACK_GRIDSEQ_SYNC_LAST] = r.transition_detector();
        m.fields[SRACK_GRIDSEQ_LAST] = f32();
        r.boolean();  // ui_dirty
        return m;
    }
    if (variant == "PatternSequencerModuleV0") {
        // sequencer.rs:336-349: id, gate_outs, sync_out, sequence, current_step, transition_detector, sync_transition_detector, ui_dirty
        r.array_of(8, "PatternSequencerModule");
        Module m = fresh(SRACK_MOD_PATTERN_SEQUENCER);
        m.id = r.str();
        if (r.array() != 8) r.fail("PatternSequencerModule: expected 8 gate outputs");
        for (int port = 0; port < 8; port++) keep_buffer(m, port, r.audio_buffer(), cfg);
        keep_buffer(m, 8, r.audio_buffer(), cfg);
        if (r.array() != 8) r.fail("PatternSequencerModule: expected 8 channels");
        uint32_t len = 0;
        for (int ch = 0; ch < 8; ch++) {
            const uint32_t n = r.array();
            if (ch == 0) len = n;
            if (n < 1 || n > 64 || n != len) r.fail("PatternSequencerModule: channel lengths must be equal and 1..64");
            for (uint32_t i = 0; i < n; i++) {
                if (r.nil()) continue;
                const bool hold = r.boolean();
                m.cells[i] |= (1u | (hold ? 2u : 0u)) << (2 * ch);
            }
        }
        m.fields[SRACK_PATSEQ_LENGTH] = len;
        m.fields[SRACK_PATSEQ_CURRENT_STEP] = (double)r.integer_in(0, 65535, "current_step");
        m.fields[SRACK_PATSEQ_STEP_LAST] = r.transition_detector();
        m.fields[SRACK_PATSEQ_SYNC_LAST] = r.transition_detector();
        r.boolean();  // ui_dirty
        return m;
    }
    if (variant == "NoiseModuleV0") {  // oscillator.rs:308-312: id, out
        r.array_of(2, "NoiseModule");
        Module m = fresh(SRACK_MOD_NOISE);
        m.id = r.str();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        return m;
    }
    if (variant == "FreeverbModuleV0") {  // freeverb.rs:8-31 minus the serde(skip) members: id, left_out, right_out, sample_rate, then
        r.array_of(16, "FreeverbModule");  // (value, ctl) pairs of dampening, freeze, wet, width, room_size, dry
        Module m = fresh(SRACK_MOD_FREEVERB);
        m.id = r.str();
        keep_buffer(m, 0, r.audio_buffer(), cfg);
        keep_buffer(m, 1, r.audio_buffer(), cfg);
        r.integer_in(0, 1000000000, "sample_rate");  // set_audio_config (freeverb.rs:126-133) replaces it with the host's; the reverb itself is never saved
        // calc() builds the reverb on the first block after a load and applies every *_ctl (set_freeverb(true)): the ctl members are the parameters
        auto pair = [&](int field) {
            r.number();
            m.fields[(size_t)field] = r.number();
        };
        pair(SRACK_FREEVERB_DAMPENING);
        r.boolean();
        m.fields[SRACK_FREEVERB_FREEZE] = r.boolean() ? 1.0 : 0.0;
        pair(SRACK_FREEVERB_WET);
        pair(SRACK_FREEVERB_WIDTH);
        pair(SRACK_FREEVERB_ROOM_SIZE);
        pair(SRACK_FREEVERB_DRY);
        return m;
    }
    std::string shown;
    for (char ch : variant.substr(0, 48)) shown += (ch >= 0x20 && ch < 0x7f) ? ch : '?';  // a damaged file: keep the message printable
    r.fail("unknown SynthModuleType variant '" + shown + "'");
}

// ---- MessagePack writer (rmp's shortest encodings) ----------------------------------------------------------------------
struct Writer {
    std::vector<uint8_t> out;
    void be(uint64_t v, int n)
    {
        for (int i = n - 1; i >= 0; i--) out.push_back((uint8_t)(v >> (8 * i)));
    }
    void nil() { out.push_back(0xc0); }
    void boolean(bool b) { out.push_back(b ? 0xc3 : 0xc2); }
    void uint(uint64_t v)
    {
        if (v < 128)
            out.push_back((uint8_t)v);
        else if (v < 256)
            out.push_back(0xcc), be(v, 1);
        else if (v < 65536)
            out.push_back(0xcd), be(v, 2);
        else if (v < (1ull << 32))
            out.push_back(0xce), be(v, 4);
        else
            out.push_back(0xcf), be(v, 8);
    }
    void f32(float f)
    {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        out.push_back(0xca);
        be(u, 4);
    }
    void f64(double d)
    {
        uint64_t u;
        std::memcpy(&u, &d, 8);
        out.push_back(0xcb);
        be(u, 8);
    }
    void str(const std::string& s)
    {
        const size_t n = s.size();
        if (n < 32)
            out.push_back((uint8_t)(0xa0 | n));
        else if (n < 256)
            out.push_back(0xd9), be(n, 1);
        else if (n < 65536)
            out.push_back(0xda), be(n, 2);
        else
            out.push_back(0xdb), be(n, 4);
        out.insert(out.end(), s.begin(), s.end());
    }
    void array(size_t n)
    {
        if (n < 16)
            out.push_back((uint8_t)(0x90 | n));
        else if (n < 65536)
            out.push_back(0xdc), be(n, 2);
        else
            out.push_back(0xdd), be(n, 4);
    }
    void variant(const char* name)
    {
        out.push_back(0x81);
        str(name);
    }
    void f32_array(const float* v, size_t n)
    {
        array(n);
        for (size_t i = 0; i < n; i++) f32(v[i]);
    }
    void transition_detector(double last)
    {
        array(1);
        boolean(last != 0.0);
    }
};

void write_buffer(Writer& w, const Module& m, int port, uint32_t B)
{
    if ((size_t)port < m.out_init.size() && m.out_init[(size_t)port].size() == (size_t)B) {
        w.f32_array(m.out_init[(size_t)port].data(), B);
    } else {  // AudioBuffer::new(Some(buffer_size)): never None — calc() unwraps it
        w.array(B);
        for (uint32_t i = 0; i < B; i++) w.f32(0.0f);
    }
}

void write_module(Writer& w, const Module& m, const AudioConfig& cfg)
{
    const uint32_t B = cfg.buffer_size;
    auto F = [&](int f) { return (float)m.fields[(size_t)f]; };
    switch (m.type) {
    case SRACK_MOD_OUTPUT:
        w.variant("OutputModuleV0");
        w.array(2);
        w.str(m.id);
        w.array((size_t)m.n_in);
        for (int c = 0; c < m.n_in; c++) write_buffer(w, m, -1, B);
        break;
    case SRACK_MOD_OSCILLATOR:
        w.variant("OscillatorModuleV0");
        w.array(9);
        w.str(m.id);
        w.f32(F(SRACK_OSC_VAL));
        w.uint(cfg.sample_rate);
        for (int port = 0; port < 3; port++) write_buffer(w, m, port, B);
        w.f64(m.fields[SRACK_OSC_POS]);
        w.boolean(m.fields[SRACK_OSC_ANTIALIASING] != 0.0);
        w.transition_detector(m.fields[SRACK_OSC_SYNC_LAST]);
        break;
    case SRACK_MOD_MOOG_FILTER:
        w.variant("MoogFilterModuleV1");
        w.array(8);
        w.str(m.id);
        for (int port = 0; port < 3; port++) write_buffer(w, m, port, B);
        w.f32(F(SRACK_VCF_FREQ));
        w.f32(F(SRACK_VCF_RES));
        w.f32(F(SRACK_VCF_EXP_AMT));
        w.array(6);
        w.f32(F(SRACK_VCF_ST_F));
        w.f32(F(SRACK_VCF_ST_P));
        w.f32(F(SRACK_VCF_ST_Q));
        w.array(5);
        for (int k = 0; k < 5; k++) w.f32(F(SRACK_VCF_ST_B0 + k));
        w.f32(F(SRACK_VCF_ST_FREQ));
        w.f32(F(SRACK_VCF_ST_RES));
        break;
    case SRACK_MOD_ADSR:
        w.variant("ADSRModuleV0");
        w.array(13);
        w.str(m.id);
        w.f32(F(SRACK_ADSR_A_SEC));
        w.f32(F(SRACK_ADSR_D_SEC));
        w.f32(F(SRACK_ADSR_S_VAL));
        w.f32(F(SRACK_ADSR_R_SEC));
        w.f32(F(SRACK_ADSR_PHASE));
        w.str(kAdsrModes[(int)m.fields[SRACK_ADSR_MODE] % 5]);
        w.f32(F(SRACK_ADSR_R_VAL));
        w.f32(F(SRACK_ADSR_FROM_A_VAL));
        w.f32(F(SRACK_ADSR_SAMPLE_RATE));
        w.transition_detector(m.fields[SRACK_ADSR_GATE_LAST]);
        write_buffer(w, m, 0, B);
        w.boolean(false);
        break;
    case SRACK_MOD_VCA:
        w.variant("VCAModuleV0");
        w.array(3);
        w.str(m.id);
        write_buffer(w, m, 0, B);
        w.boolean(m.fields[SRACK_VCA_NEGATIVE] != 0.0);
        break;
    case SRACK_MOD_MONO_MIXER:
        w.variant("MonoMixerModuleV0");
        w.array(3);
        w.str(m.id);
        w.array((size_t)m.n_in);
        for (int k = 0; k < m.n_in; k++) w.f32(F(SRACK_MIX_GAIN0 + k));
        write_buffer(w, m, 0, B);
        break;
    case SRACK_MOD_MATH:
        w.variant("MathModuleV0");
        w.array(4);
        w.str(m.id);
        write_buffer(w, m, 0, B);
        w.f32(F(SRACK_MATH_CONSTANT));
        w.str(kMathOps[(int)m.fields[SRACK_MATH_OPERATION] % 3]);
        break;
    case SRACK_MOD_NONLINEAR:
        w.variant("NonLinearModuleV0");
        w.array(3);
        w.str(m.id);
        write_buffer(w, m, 0, B);
        w.f32(F(SRACK_NONLIN_CONSTANT));
        break;
    case SRACK_MOD_FREEVERB:
        w.variant("FreeverbModuleV0");
        w.array(16);
        w.str(m.id);
        write_buffer(w, m, 0, B);
        write_buffer(w, m, 1, B);
        w.uint(cfg.sample_rate);
        for (int f : {SRACK_FREEVERB_DAMPENING, SRACK_FREEVERB_FREEZE, SRACK_FREEVERB_WET, SRACK_FREEVERB_WIDTH, SRACK_FREEVERB_ROOM_SIZE, SRACK_FREEVERB_DRY})
            for (int twice = 0; twice < 2; twice++) {  // the applied value and the slider's: equal once a block has run
                if (f == SRACK_FREEVERB_FREEZE)
                    w.boolean(m.fields[(size_t)f] != 0.0);
                else
                    w.f64(m.fields[(size_t)f]);
            }
        break;
    case SRACK_MOD_NOISE:
        w.variant("NoiseModuleV0");
        w.array(2);
        w.str(m.id);
        write_buffer(w, m, 0, B);
        break;
    case SRACK_MOD_SAMPLE:
        w.variant("SampleModuleV0");
        w.array(7);
        w.str(m.id);
        w.transition_detector(m.fields[SRACK_SAMPLE_GATE_LAST]);
        w.f32(F(SRACK_SAMPLE_POS));
        write_buffer(w, m, 0, B);
        w.array(3);
        w.f32_array(m.wave.data(), m.wave.size());
        w.f32(F(SRACK_SAMPLE_WAVE_SAMPLE_RATE));
        w.boolean(m.fields[SRACK_SAMPLE_WAVE_NEW] != 0.0);
        w.boolean(m.fields[SRACK_SAMPLE_PLAYING] != 0.0);
        w.f32(F(SRACK_SAMPLE_SAMPLE_RATE));
        break;
    case SRACK_MOD_GRID_SEQUENCER: {
        w.variant("GridSequencerModuleV1");
        w.array(12);
        w.str(m.id);
        for (int port = 0; port < 3; port++) write_buffer(w, m, port, B);
        const int len = (int)m.fields[SRACK_GRIDSEQ_LENGTH];
        w.array((size_t)len);
        for (int i = 0; i < len; i++) {
            const uint32_t cell = m.cells[(size_t)i];
            if (!(cell & 0x80000000u)) {
                w.nil();
            } else {
                w.array(2);
                w.uint(cell & 0xffffu);
                w.boolean(cell & 0x40000000u);
            }
        }
        w.uint((uint64_t)m.fields[SRACK_GRIDSEQ_OCTAVES]);
        w.uint((uint64_t)m.fields[SRACK_GRIDSEQ_STEPS_PER_OCTAVE]);
        w.uint((uint64_t)m.fields[SRACK_GRIDSEQ_CURRENT_STEP]);
        w.transition_detector(m.fields[SRACK_GRIDSEQ_STEP_LAST]);
        w.transition_detector(m.fields[SRACK_GRIDSEQ_SYNC_LAST]);
        w.f32(F(SRACK_GRIDSEQ_LAST));
        w.boolean(false);
        break;
    }
    case SRACK_MOD_PATTERN_SEQUENCER: {
        w.variant("PatternSequencerModuleV0");
        w.array(8);
        w.str(m.id);
        w.array(8);
        for (int port = 0; port < 8; port++) write_buffer(w, m, port, B);
        write_buffer(w, m, 8, B);
        const int len = (int)m.fields[SRACK_PATSEQ_LENGTH];
        w.array(8);
        for (int ch = 0; ch < 8; ch++) {
            w.array((size_t)len);
            for (int i = 0; i < len; i++) {
                const uint32_t b = (m.cells[(size_t)i] >> (2 * ch)) & 3u;
                if (!(b & 1u))
                    w.nil();
                else
                    w.boolean(b & 2u);
            }
        }
        w.uint((uint64_t)m.fields[SRACK_PATSEQ_CURRENT_STEP]);
        w.transition_detector(m.fields[SRACK_PATSEQ_STEP_LAST]);
        w.transition_detector(m.fields[SRACK_PATSEQ_SYNC_LAST]);
        w.boolean(false);
        break;
    }
    }
}

}  // namespace

// FileFormat -> Graph, the way SynthModuleWorkspaceImpl::deserialize does it (ui.rs:116-135).  `g.cfg` is the host's
// AudioConfig; g.modules is replaced.
int load_srk(const uint8_t* bytes, size_t n_bytes, Graph& g)
{
    try {
        Reader r{bytes, bytes + n_bytes};
        Graph scratch;
        scratch.cfg = g.cfg;
        r.array_of(3, "FileFormat");
        std::vector<Module> file_modules;
        for (uint32_t i = 0, n = r.array(); i < n; i++) file_modules.push_back(parse_module(r, scratch));
        struct Conn {
            std::string src, sink;
            int src_port, sink_port;
        };
        std::vector<Conn> conns;
        for (uint32_t i = 0, n = r.array(); i < n; i++) {
            r.array_of(4, "connection");
            Conn c;
            c.src = r.str();
            c.src_port = (int)r.integer_in(0, 255, "src_port");
            c.sink = r.str();
            c.sink_port = (int)r.integer_in(0, 255, "sink_port");
            conns.push_back(std::move(c));
        }
        std::map<std::string, std::pair<float, float>> positions;
        for (uint32_t i = 0, n = r.array(); i < n; i++) {
            r.array_of(2, "position");
            std::string id = r.str();
            r.array_of(2, "(f32, f32)");
            const float x = (float)r.number(), y = (float)r.number();
            positions.emplace(std::move(id), std::make_pair(x, y));  // popped from the end, inserted over: the first entry of an id wins
        }
        if (r.p != r.end) r.fail("trailing bytes after FileFormat");

        g.modules.clear();
        for (size_t i = file_modules.size(); i-- > 0;) g.modules.push_back(std::move(file_modules[i]));  // unpack_modules pops (ui.rs:654-660)
        std::map<std::string, int> by_id;
        for (size_t i = 0; i < g.modules.size(); i++) {
            Module& m = g.modules[i];
            by_id[m.id] = (int)i;
            auto pos = positions.find(m.id);
            if (pos != positions.end()) {
                m.has_pos = true;
                m.pos_x = pos->second.first;
                m.pos_y = pos->second.second;
            }
        }
        for (size_t k = conns.size(); k-- > 0;) {  // unpack_connections pops; unknown ids and bad ports are dropped (ui.rs:672-680)
            auto sink = by_id.find(conns[k].sink), src = by_id.find(conns[k].src);
            if (sink == by_id.end() || src == by_id.end()) continue;
            (void)g.connect(src->second, conns[k].src_port, sink->second, conns[k].sink_port);
        }
        g.plan.valid = false;
        g.revision++;
        return SRACK_OK;
    } catch (const UnsupportedError& e) {
        set_error(e.what());
        return SRACK_ERR_UNSUPPORTED;
    } catch (const ParseError& e) {
        set_error(e.what());
        return SRACK_ERR_INVALID;
    } catch (const std::bad_alloc&) {
        set_error("srk: out of memory");
        return SRACK_ERR_NOMEM;
    }
}

// Graph -> FileFormat, like SynthModuleWorkspaceImpl::serialize (ui.rs:98-114): modules in workspace order, the
// connections of every module's inputs in port order (ui.rs:612-637), positions of the modules that have one.
std::vector<uint8_t> save_srk(const Graph& g)
{
    Writer w;
    w.array(3);
    w.array(g.modules.size());
    for (const Module& m : g.modules) write_module(w, m, g.cfg);
    size_t n_conn = 0, n_pos = 0;
    for (const Module& m : g.modules) {
        for (const InputRef& in : m.in) n_conn += in.src >= 0;
        n_pos += m.has_pos;
    }
    w.array(n_conn);
    for (const Module& m : g.modules)
        for (size_t k = 0; k < m.in.size(); k++) {
            if (m.in[k].src < 0) continue;
            w.array(4);
            w.str(g.modules[(size_t)m.in[k].src].id);
            w.uint((uint64_t)m.in[k].port);
            w.str(m.id);
            w.uint(k);
        }
    w.array(n_pos);
    for (const Module& m : g.modules) {
        if (!m.has_pos) continue;
        w.array(2);
        w.str(m.id);
        w.array(2);
        w.f32(m.pos_x);
        w.f32(m.pos_y);
    }
    return std::move(w.out);
}

}  // namespace srack
