// srack.hpp — C++ host-side mirror of the reference's module-graph API, on top of the C ABI (srack_hip.h).
//
// The reference is Rust; its toolchain is not part of this build, so the host side above the C ABI is
// given in C++ with the reference's names and error behaviour, so that code (and tests) written against
// `trait SynthModule` / `plan_execution` / `execute` (src/synth.rs:97-263) read the same here:
//
//   reference                                   this header
//   ------------------------------------------  -------------------------------------------------------
//   AudioConfig{sample_rate, buffer_size, ch}    srack::AudioConfig
//   Arc<RwLock<dyn SynthModule>> (SharedSynthModule)  srack::SharedSynthModule (value handle: workspace + index)
//   OscillatorModule::new(&cfg) pushed to the    ws.add(srack::ModuleType::Oscillator)
//     workspace's module list (ui.rs:54)
//   m.set_input(i, src, port) -> Result<(),()>   m.set_input(i, src, port) -> bool   (false = Err(()))
//   m.get_input(i) / disconnect_input(i)         m.get_input(i) / m.disconnect_input(i)
//   m.get_num_inputs() / get_num_outputs()       same
//   plan_execution(output, &all_modules, &mut plan)   srack::plan_execution(output, all_modules, plan)
//   execute(&plan) once per buffer_size frames   ws.execute_batch(n_voices, n_samples, d_frames, d_mix)
//
// Header-only; link with libsrack_hip.so.  No HIP types appear: device buffers are void* / float*.
#pragma once
#include <cstdint>
#include <array>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "srack_hip.h"

namespace srack {

struct AudioConfig {  // synth.rs:20-25
    uint16_t sample_rate = 48000;
    size_t buffer_size = 1024;
    uint8_t channels = 2;
};

enum class ModuleType : int {  // the hot-path subset of SynthModuleType (synth.rs:300-317)
    Output = SRACK_MOD_OUTPUT,
    Oscillator = SRACK_MOD_OSCILLATOR,
    MoogFilter = SRACK_MOD_MOOG_FILTER,
    ADSR = SRACK_MOD_ADSR,
    VCA = SRACK_MOD_VCA,
    MonoMixer = SRACK_MOD_MONO_MIXER,
    Math = SRACK_MOD_MATH,
    GridSequencer = SRACK_MOD_GRID_SEQUENCER,
    PatternSequencer = SRACK_MOD_PATTERN_SEQUENCER,
    NonLinear = SRACK_MOD_NONLINEAR,
    Sample = SRACK_MOD_SAMPLE,
    Noise = SRACK_MOD_NOISE,
    Freeverb = SRACK_MOD_FREEVERB
};

class Error : public std::runtime_error {  // what the reference would panic with
public:
    Error(int code, const std::string& what) : std::runtime_error(what), code(code) {}
    int code;
};

class Workspace;

// Value handle with pointer-identity semantics, like an Arc clone (synth.rs:270, ByAddress in the planner).
class SharedSynthModule {
public:
    SharedSynthModule() = default;
    bool operator==(const SharedSynthModule& o) const { return ws_ == o.ws_ && index_ == o.index_; }
    bool operator!=(const SharedSynthModule& o) const { return !(*this == o); }
    int index() const { return index_; }

    std::string get_name() const;  // SynthModule::get_name
    uint8_t get_num_inputs() const;
    uint8_t get_num_outputs() const;
    // Result<(), ()>: false on a bad port index
    bool set_input(uint8_t input_idx, const SharedSynthModule& src_module, uint8_t src_port);
    bool disconnect_input(uint8_t input_idx);
    void disconnect_inputs()
    {
        for (uint8_t i = 0; i < get_num_inputs(); i++) disconnect_input(i);
    }
    // Result<Option<(SharedSynthModule, u8)>, ()>: outer nullopt = Err(()), inner nullopt = None
    std::optional<std::optional<std::pair<SharedSynthModule, uint8_t>>> get_input(uint8_t input_idx) const;
    // struct fields (what the egui sliders and serde touch), by the SRACK_<TYPE>_<FIELD> enums
    void set(int field, double value);
    double get(int field) const;
    // sequencer grid cells (sequencer.rs:137-184, 437-478) and the sample player's WaveBox (sample.rs:31-69)
    void set_step(int channel, int step, int state, int value = 0);
    void set_wave(const std::vector<float>& samples, float sample_rate);

private:
    friend class Workspace;
    SharedSynthModule(Workspace* ws, int index) : ws_(ws), index_(index) {}
    Workspace* ws_ = nullptr;
    int index_ = -1;
};

// SynthModuleWorkspaceImpl (ui.rs:52-60): owns the module list and the plan.
class Workspace {
public:
    explicit Workspace(const AudioConfig& cfg) : cfg_(cfg)
    {
        int rc = srack_patch_create(cfg.sample_rate, (uint32_t)cfg.buffer_size, cfg.channels, &p_);
        if (rc != SRACK_OK) throw Error(rc, srack_last_error());
    }
    ~Workspace() { srack_patch_destroy(p_); }
    Workspace(const Workspace&) = delete;
    Workspace& operator=(const Workspace&) = delete;

    // Module::new(&audio_config) + push to `modules`
    SharedSynthModule add(ModuleType t)
    {
        int i = srack_patch_add_module(p_, (int)t);
        if (i < 0) throw Error(i, srack_last_error());
        return SharedSynthModule(this, i);
    }
    std::vector<SharedSynthModule> modules()
    {
        std::vector<SharedSynthModule> v;
        for (int i = 0; i < srack_patch_num_modules(p_); i++) v.emplace_back(SharedSynthModule(this, i));
        return v;
    }
    // SynthModuleWorkspaceImpl::plan (ui.rs:63-82): output = first OutputModule; empty plan when there is none
    std::vector<SharedSynthModule> plan()
    {
        int order[1024];
        int n = srack_patch_plan(p_, order, 1024);
        std::vector<SharedSynthModule> v;
        for (int i = 0; i < n; i++) v.emplace_back(SharedSynthModule(this, order[i]));
        return v;
    }
    // the wires that became buffer_size-sample delays (src, src_port, sink, sink_port)
    std::vector<std::array<int, 4>> delayed_edges()
    {
        int q[4 * 256];
        int n = srack_patch_delayed_edges(p_, q, 256);
        std::vector<std::array<int, 4>> v;
        for (int i = 0; i < n; i++) v.push_back({q[4 * i], q[4 * i + 1], q[4 * i + 2], q[4 * i + 3]});
        return v;
    }

    // ---- the batch counterpart of `execute(&plan)` ---------------------------------------------------
    void configure_voices(uint32_t n_voices) { check(srack_voices_configure(p_, n_voices)); }
    // the reference's sliders and cables change under a running graph: edits between execute_batch calls keep the voices' state
    void keep_state(bool keep = true) { check(srack_patch_keep_state(p_, keep ? 1 : 0)); }
    // NoiseModule streams: (seed, module, first_voice + voice); the reference's generator is OS-seeded
    void set_noise_seed(uint64_t seed, uint64_t first_voice = 0) { check(srack_patch_set_noise_seed(p_, seed, first_voice)); }
    void set_voice_field(const SharedSynthModule& m, int field, const float* values) { check(srack_voices_set_field_f32(p_, m.index(), field, values)); }
    int planes(int* channel_plane = nullptr, int cap = 0) { return check(srack_render_planes(p_, channel_plane, cap)); }
    // n_samples ticks for every voice, continuing from the current state; device pointers, asynchronous on `stream`
    void execute_batch(uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags = 0, void* stream = nullptr)
    {
        check(srack_render(p_, n_samples, d_frames, d_mix, flags, stream));
    }

    srack_patch* handle() { return p_; }
    const AudioConfig& config() const { return cfg_; }
    int check(int rc) const
    {
        if (rc < 0) throw Error(rc, srack_last_error());
        return rc;
    }

private:
    friend class SharedSynthModule;
    AudioConfig cfg_;
    srack_patch* p_ = nullptr;
};

// The multi-GPU half (no reference counterpart: s-rack is single-threaded, main.rs:59-63): one process per GPU, voices sharded by
// global voice index, and one collective — the sum of the per-rank partial mixes.  Rank 0 makes the id, the host carries its
// 128 bytes to every rank, every rank constructs a MixComm with its device already selected (srack_device_set).
class MixComm {
public:
    static std::string unique_id()
    {
        std::string id((size_t)SRACK_DIST_ID_BYTES, '\0');
        const int rc = srack_dist_unique_id(&id[0]);
        if (rc < 0) throw Error(rc, srack_last_error());
        return id;
    }
    MixComm(const std::string& id, int n_ranks, int rank)
    {
        if (id.size() != (size_t)SRACK_DIST_ID_BYTES) throw Error(SRACK_ERR_INVALID, "MixComm: the id is SRACK_DIST_ID_BYTES bytes");
        const int rc = srack_dist_init(id.data(), n_ranks, rank, &comm_);
        if (rc < 0) throw Error(rc, srack_last_error());
    }
    MixComm(const MixComm&) = delete;
    MixComm& operator=(const MixComm&) = delete;
    ~MixComm() { srack_dist_destroy(comm_); }
    int count() const
    {
        int n = 0;
        const int rc = srack_dist_comm_count(comm_, &n);
        if (rc < 0) throw Error(rc, srack_last_error());
        return n;
    }
    // ncclReduce(sum, f32) of `count` floats at device pointer d_mix, in place, asynchronous on `stream`
    void reduce_mix(float* d_mix, size_t count, int root = 0, void* stream = nullptr)
    {
        const int rc = srack_dist_reduce_mix(comm_, d_mix, count, root, stream);
        if (rc < 0) throw Error(rc, srack_last_error());
    }

private:
    void* comm_ = nullptr;
};

inline std::string SharedSynthModule::get_name() const
{
    static const char* names[] = {"Output", "Oscillator", "Moog Filter", "ADSR", "VCA", "Mono Mixer", "Math",
                                  "Grid Sequencer", "Pattern Sequencer", "Non-Linear", "Sample", "Noise", "Freeverb"};  // each module's get_name()
    int t = srack_patch_module_type(ws_->p_, index_);
    if (t == SRACK_MOD_MATH) {
        static const char* ops[] = {"Add", "Subtract", "Multiply"};  // math.rs:37-43
        return ops[(int)get(SRACK_MATH_OPERATION) % 3];
    }
    return t >= 0 && t < SRACK_MOD__COUNT ? names[t] : "?";
}
inline uint8_t SharedSynthModule::get_num_inputs() const { return (uint8_t)ws_->check(srack_module_num_inputs(ws_->p_, index_)); }
inline uint8_t SharedSynthModule::get_num_outputs() const { return (uint8_t)ws_->check(srack_module_num_outputs(ws_->p_, index_)); }
inline bool SharedSynthModule::set_input(uint8_t input_idx, const SharedSynthModule& src, uint8_t src_port)
{
    return srack_patch_connect(ws_->p_, src.index_, src_port, index_, input_idx) == SRACK_OK;
}
inline bool SharedSynthModule::disconnect_input(uint8_t input_idx) { return srack_patch_disconnect(ws_->p_, index_, input_idx) == SRACK_OK; }
inline std::optional<std::optional<std::pair<SharedSynthModule, uint8_t>>> SharedSynthModule::get_input(uint8_t input_idx) const
{
    int m = -1, port = 0;
    if (srack_patch_get_input(ws_->p_, index_, input_idx, &m, &port) != SRACK_OK) return std::nullopt;  // Err(())
    if (m < 0) return std::optional<std::pair<SharedSynthModule, uint8_t>>{};                            // Ok(None)
    return std::optional<std::pair<SharedSynthModule, uint8_t>>{std::make_pair(SharedSynthModule(ws_, m), (uint8_t)port)};
}
inline void SharedSynthModule::set(int field, double value) { ws_->check(srack_patch_set_field(ws_->p_, index_, field, value)); }
inline void SharedSynthModule::set_step(int channel, int step, int state, int value) { ws_->check(srack_patch_set_step(ws_->p_, index_, channel, step, state, value)); }
inline void SharedSynthModule::set_wave(const std::vector<float>& samples, float sample_rate)
{
    ws_->check(srack_patch_set_wave(ws_->p_, index_, samples.data(), (uint32_t)samples.size(), sample_rate));
}
inline double SharedSynthModule::get(int field) const
{
    double v = 0;
    ws_->check(srack_patch_get_field(ws_->p_, index_, field, &v));
    return v;
}

// get_inputs (synth.rs:214-218)
inline std::vector<std::optional<std::pair<SharedSynthModule, uint8_t>>> get_inputs(const SharedSynthModule& m)
{
    std::vector<std::optional<std::pair<SharedSynthModule, uint8_t>>> v;
    for (uint8_t i = 0; i < m.get_num_inputs(); i++) v.push_back(*m.get_input(i));
    return v;
}

// plan_execution(output, &all_modules, &mut plan) (synth.rs:128-212), explicit list as in the reference's test
inline void plan_execution(Workspace& ws, const SharedSynthModule& output, const std::vector<SharedSynthModule>& all_modules,
                           std::vector<SharedSynthModule>& plan)
{
    std::vector<int> list, order(all_modules.size() + 1);
    for (const auto& m : all_modules) list.push_back(m.index());
    int n = ws.check(srack_patch_plan_list(ws.handle(), output.index(), list.data(), (int)list.size(), order.data(), (int)order.size()));
    auto mods = ws.modules();
    plan.clear();
    for (int i = 0; i < n; i++) plan.push_back(mods[(size_t)order[(size_t)i]]);
}

}  // namespace srack
