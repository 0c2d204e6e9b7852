// graph.hpp — host-side mirror of the reference's workspace graph for the batch-render path.
//
// What it mirrors (all /root/reference/src):
//   * the module list `all_modules` and per-module input slots     ui.rs:54, e.g. vca.rs:10-12
//   * SynthModule::{get_num_inputs,get_num_outputs,set_input,disconnect_input,get_input}
//                                                                  synth.rs:222-263
//   * Module::new(&AudioConfig) defaults                           oscillator.rs:27-41, filter.rs:28-41,
//                                                                  adsr.rs:36-53, vca.rs:18-26, mixer.rs:16-23,
//                                                                  math.rs:26-35, output.rs:15-23
//   * plan_execution / is_loop / get_inputs                        synth.rs:107-218
// Modules are value types addressed by their index in the list (the reference uses Arc pointer
// identity, synth.rs:109); fields are stored as doubles (exact for f32 / f64 / bool / enum).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/srack_hip.h"

namespace srack {

struct InputRef {
    int src = -1;  // module index, -1 = None
    int port = 0;
};

struct Module {
    int type = -1;
    int n_in = 0, n_out = 0;
    std::vector<InputRef> in;
    std::vector<double> fields;  // indexed by the SRACK_<TYPE>_* field enums
    // sequencers: 64 grid cells in device format.  Grid: bit 31 present, bit 30 hold, bits 0..15 note value;
    // pattern: bits 2c / 2c+1 = present / hold of channel c.
    std::vector<uint32_t> cells;
    std::vector<float> wave;  // SampleModule: wavebox.samples
    uint64_t wave_revision = 0;  // Graph::revision when the wave (and with it wavebox.new) was last set
    // what a .srk file carries besides the fields (ui.rs:578-586): the module's UUID string, its workspace position,
    // and the contents of its output buffers — the latter is what the sink of a broken feedback edge reads during
    // the first block after a load (empty = zeros, AudioBuffer::new)
    std::string id;
    bool has_pos = false;
    float pos_x = 0.0f, pos_y = 0.0f;
    std::vector<std::vector<float>> out_init;  // per output port: buffer_size samples, or empty
};

struct AudioConfig {  // synth.rs:20-25
    uint32_t sample_rate = 48000;
    uint32_t buffer_size = 1024;
    uint32_t channels = 2;
    // not in the reference (its NoiseModule draws from an OS-seeded thread-local generator): srack_patch_set_noise_seed
    uint64_t noise_seed = 0, noise_first_voice = 0;
};

uint64_t splitmix64(uint64_t x);                         // output function applied to x + 0x9E3779B97F4A7C15
uint64_t noise_base_key(uint64_t seed, int module);      // sm(seed ^ sm(module)), see srack_hip.h

struct Edge {  // a wire src.port -> sink.port
    int src, src_port, sink, sink_port;
};

struct Plan {
    bool valid = false;
    int output = -1;                // the OutputModule found by find_output (ui.rs:84-96), -1 if none
    std::vector<int> order;         // execution order (module indices)
    std::vector<int> position;      // position[module] in `order`, -1 if unscheduled
    std::vector<std::pair<int, int>> removed;  // (from, module): scheduler edges dropped by phase 2
};

class Graph {
public:
    AudioConfig cfg;
    std::vector<Module> modules;
    Plan plan;
    uint64_t revision = 0;  // bumped by every structural or field edit

    int add_module(int type);  // returns index or SRACK_ERR_*
    int num_fields(int module) const;
    int set_field(int module, int field, double value);
    int get_field(int module, int field, double* value) const;
    int set_step(int module, int channel, int step, int state, int value);
    int get_step(int module, int channel, int step, int* state, int* value) const;
    int set_wave(int module, const float* samples, uint32_t n, float sample_rate);
    int set_output_buffer(int module, int port, const float* samples, uint32_t n);  // n == buffer_size, or 0 to clear
    int connect(int src, int src_port, int sink, int sink_port);
    int disconnect(int sink, int sink_port);

    // plan_execution driven like SynthModuleWorkspaceImpl::plan (ui.rs:63-82)
    int make_plan();
    // plan_execution(output, all_modules, plan) with an explicit list (the reference test shuffles it)
    int make_plan(int output, const std::vector<int>& all_modules);

    // wires whose source runs AFTER its sink in the plan: the sink reads the previous block's
    // buffer => a buffer_size-sample delay
    std::vector<Edge> delayed_edges() const;

    static int fields_of_type(int type);
    static bool field_is_state(int type, int field);
    static bool field_is_f64(int type, int field);
    static bool field_is_flag(int type, int field);  // bool / enum stored as integer
};

std::string new_module_id();  // uuid::Uuid::new_v4().to_string() stand-in (unique per process, version-4 layout)

void set_error(const std::string& msg);
const char* last_error();

}  // namespace srack
