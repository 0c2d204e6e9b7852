// runtime.hpp — the object behind `srack_patch*`: graph + voices + flattened program + device state.
#pragma once
#include <cstdint>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "flatten.hpp"
#include "graph.hpp"

namespace srack {

struct DeviceState;  // HIP side, render.hip

struct PatchHandle {
    Graph graph;
    uint32_t n_voices = 0;
    std::vector<VoiceOverride> overrides;
    uint64_t voices_revision = 0;  // bumped by configure / per-voice set_field

    FlatPair prog;
    bool prog_valid = false;
    uint64_t prog_graph_revision = 0, prog_voices_revision = 0;
    uint32_t prog_flags = 0;

    DeviceState* dev = nullptr;
    uint64_t samples_rendered = 0;  // absolute tick count: phase of the feedback rings
    bool keep_state = false;        // srack_patch_keep_state: carry the modules' device state across a re-flatten
    bool voices_fresh = true;       // set by srack_voices_configure: nothing on the device belongs to these voices yet
    // keep_state: the device state of the program that was replaced, kept until the new one is uploaded (rings and reverb lines
    // move device to device), with what it held and where
    DeviceState* dev_old = nullptr;
    struct OldTag {
        int stage;  // -1: the voice program
        uint32_t n_voices;
        FlatProgram::CarryTag tag;
    };
    std::vector<OldTag> old_tags;
    // keep_state: (module, field) pairs of STATE fields the host wrote since the last flatten — the carry leaves those alone
    // (an explicit srack_patch_set_field / srack_voices_set_field_* on a state field wins over the running value)
    std::set<std::pair<int, int>> state_writes;
    bool timing_armed = false;      // srack_render_kernel_ms has been called: renders bracket the dominant kernel with HIP events

    ~PatchHandle();
};

// (Re)flatten when the graph, the voices or the flags changed since the last render; a re-flatten
// resets the device voice state to the modules' fields (like re-loading the patch).
int ensure_program(PatchHandle& h, uint32_t flags);
// The program `flags` would render, without touching the handle (its own if current, else flattened from a copy into `scratch`).
int peek_program(PatchHandle& h, uint32_t flags, FlatPair& scratch, const FlatPair** out);

int device_render(PatchHandle& h, uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags, void* stream);
int device_reserve(PatchHandle& h, uint32_t n_samples, bool want_mix, uint32_t flags);
int device_kernel_ms(PatchHandle& h, double* avg_ms, int* n_launches, int reset);
int device_read_rows(PatchHandle& h, int ctl_stage /* -1: the voice program */, int first_row, int n_rows, uint32_t* host_dst);
// One state field of one module as the CURRENT program holds it on the device, per voice (no re-flatten); false: not device state.
bool read_device_state(PatchHandle& h, int module, int field, std::vector<double>& values);
void device_release(DeviceState* d);
const char* device_kernel_name(const PatchHandle& h);
std::string device_jit_note(const PatchHandle& h);  // " jit=compiled(1834 ms)" / " jit=disk-cache" / " jit=memory-cache" / " jit=unavailable(why)" / ""

}  // namespace srack
