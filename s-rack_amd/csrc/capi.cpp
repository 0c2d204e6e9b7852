// capi.cpp — the extern "C" boundary declared in include/srack_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "jit.hpp"
#include "runtime.hpp"

using namespace srack;

namespace srack {  // srk.cpp
int load_srk(const uint8_t* bytes, size_t n_bytes, Graph& g);
std::vector<uint8_t> save_srk(const Graph& g);
}

struct srack_patch {
    PatchHandle h;
};

#define CHECK_HANDLE(p)                       \
    do {                                      \
        if (!(p)) {                           \
            set_error("null patch handle");   \
            return SRACK_ERR_INVALID;         \
        }                                     \
    } while (0)

#define HIP_TRY_C(expr)                                                     \
    do {                                                                    \
        hipError_t e_ = (expr);                                             \
        if (e_ != hipSuccess) {                                             \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));   \
            return SRACK_ERR_DEVICE;                                        \
        }                                                                   \
    } while (0)

// No C++ exception may cross the C boundary (a Rust or ctypes host cannot unwind it): allocation failures become
// SRACK_ERR_NOMEM, anything else SRACK_ERR_INVALID with the text in srack_last_error().
template <class F>
static int guarded(F&& f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        try { set_error("out of memory"); } catch (...) {}
        return SRACK_ERR_NOMEM;
    } catch (const std::exception& e) {
        try { set_error(std::string("internal error: ") + e.what()); } catch (...) {}
        return SRACK_ERR_INVALID;
    } catch (...) {
        try { set_error("internal error"); } catch (...) {}
        return SRACK_ERR_INVALID;
    }
}

// `(int)x` of a NaN or of a value outside int's range is undefined: flags and enum-like fields are clamped first
static double as_flag(double x)
{
    if (!(x == x)) return 0.0;
    if (x > 2147483647.0) return 2147483647.0;
    if (x < -2147483648.0) return -2147483648.0;
    return (double)(int)x;
}

template <typename T>
static int set_voice_field_impl(srack_patch* p, int module, int field, const T* values)
{
    CHECK_HANDLE(p);
    PatchHandle& h = p->h;
    if (h.n_voices == 0) {
        set_error("voices_set_field: call srack_voices_configure first");
        return SRACK_ERR_STATE;
    }
    int nf = h.graph.num_fields(module);
    if (nf < 0 || field < 0 || field >= nf || !values) {
        set_error("voices_set_field: no such module/field");
        return SRACK_ERR_INVALID;
    }
    VoiceOverride o;
    o.module = module;
    o.field = field;
    o.values.resize(h.n_voices);
    const int type = h.graph.modules[(size_t)module].type;
    const bool f64 = Graph::field_is_f64(type, field), flag = Graph::field_is_flag(type, field);
    for (uint32_t v = 0; v < h.n_voices; v++) {
        double x = (double)values[v];
        o.values[v] = f64 ? x : (flag ? as_flag(x) : (double)(float)x);
    }
    for (auto it = h.overrides.begin(); it != h.overrides.end();)
        it = (it->module == module && it->field == field) ? h.overrides.erase(it) : it + 1;
    h.overrides.push_back(std::move(o));
    h.voices_revision++;
    if (Graph::field_is_state(type, field)) h.state_writes.insert({module, field});  // keep_state: the host's value wins over the carried one
    return SRACK_OK;
}

template <typename T>
static int set_voice_field(srack_patch* p, int module, int field, const T* values)
{
    return guarded([&]() -> int { return set_voice_field_impl(p, module, field, values); });
}

extern "C" {

int srack_abi_version(void) { return SRACK_ABI_VERSION; }
const char* srack_last_error(void) { return last_error(); }

int srack_patch_create(uint32_t sample_rate, uint32_t buffer_size, uint32_t channels, srack_patch** out)
{
    return guarded([&]() -> int {
        if (!out) {
            set_error("srack_patch_create: out is null");
            return SRACK_ERR_INVALID;
        }
        *out = nullptr;
        if (sample_rate == 0 || sample_rate > 65535u) {  // AudioConfig.sample_rate is a u16 (synth.rs:22)
            set_error("srack_patch_create: sample_rate must fit the reference's u16 (1..65535)");
            return SRACK_ERR_INVALID;
        }
        if (buffer_size == 0) {
            set_error("srack_patch_create: buffer_size must be >= 1");
            return SRACK_ERR_INVALID;
        }
        if (channels == 0 || channels > 8) {
            set_error("srack_patch_create: channels must be 1..8");
            return SRACK_ERR_INVALID;
        }
        auto* p = new (std::nothrow) srack_patch();
        if (!p) return SRACK_ERR_NOMEM;
        p->h.graph.cfg = AudioConfig{sample_rate, buffer_size, channels};
        *out = p;
        return SRACK_OK;
    });
}

int srack_patch_destroy(srack_patch* p)
{
    return guarded([&]() -> int {
        delete p;
        return SRACK_OK;
    });
}

int srack_patch_add_module(srack_patch* p, int module_type)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.add_module(module_type);
    });
}

int srack_patch_num_modules(const srack_patch* p)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return (int)p->h.graph.modules.size();
    });
}

static const Module* get_module(const srack_patch* p, int module)
{
    if (!p || module < 0 || module >= (int)p->h.graph.modules.size()) {
        set_error("no such module");
        return nullptr;
    }
    return &p->h.graph.modules[(size_t)module];
}

int srack_patch_module_type(const srack_patch* p, int module)
{
    return guarded([&]() -> int {
        const Module* m = get_module(p, module);
        return m ? m->type : SRACK_ERR_INVALID;
    });
}

int srack_module_num_inputs(const srack_patch* p, int module)
{
    return guarded([&]() -> int {
        const Module* m = get_module(p, module);
        return m ? m->n_in : SRACK_ERR_INVALID;
    });
}

int srack_module_num_outputs(const srack_patch* p, int module)
{
    return guarded([&]() -> int {
        const Module* m = get_module(p, module);
        return m ? m->n_out : SRACK_ERR_INVALID;
    });
}

int srack_patch_set_field(srack_patch* p, int module, int field, double value)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const int rc = p->h.graph.set_field(module, field, value);
        if (rc == SRACK_OK && Graph::field_is_state(p->h.graph.modules[(size_t)module].type, field)) {
            // keep_state: the host's value wins over the running one — also over the per-voice values an earlier carry left behind
            PatchHandle& h = p->h;
            h.state_writes.insert({module, field});
            if (h.keep_state)
                for (auto it = h.overrides.begin(); it != h.overrides.end();) it = (it->module == module && it->field == field) ? h.overrides.erase(it) : it + 1;
        }
        return rc;
    });
}

int srack_patch_get_field(const srack_patch* p, int module, int field, double* value)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.get_field(module, field, value);
    });
}

int srack_patch_set_step(srack_patch* p, int module, int channel, int step, int state, int value)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.set_step(module, channel, step, state, value);
    });
}

int srack_patch_get_step(const srack_patch* p, int module, int channel, int step, int* state, int* value)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.get_step(module, channel, step, state, value);
    });
}

int srack_patch_set_wave(srack_patch* p, int module, const float* samples, uint32_t n_samples, float sample_rate)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.set_wave(module, samples, n_samples, sample_rate);
    });
}

int srack_patch_get_wave(const srack_patch* p, int module, float* samples, uint32_t cap, float* sample_rate)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const Module* m = get_module(p, module);
        if (!m || m->type != SRACK_MOD_SAMPLE) {
            set_error("get_wave: not a SampleModule");
            return SRACK_ERR_INVALID;
        }
        if (samples)
            for (size_t i = 0; i < m->wave.size() && i < (size_t)cap; i++) samples[i] = m->wave[i];
        if (sample_rate) *sample_rate = (float)m->fields[SRACK_SAMPLE_WAVE_SAMPLE_RATE];
        return (int)m->wave.size();
    });
}

int srack_patch_load_srk(const void* bytes, size_t n_bytes, uint32_t sample_rate, uint32_t buffer_size, uint32_t channels, srack_patch** out)
{
    return guarded([&]() -> int {
        if (!bytes && n_bytes) {
            set_error("srack_patch_load_srk: bytes is null");
            return SRACK_ERR_INVALID;
        }
        int rc = srack_patch_create(sample_rate, buffer_size, channels, out);
        if (rc != SRACK_OK) return rc;
        rc = load_srk((const uint8_t*)bytes, n_bytes, (*out)->h.graph);
        if (rc != SRACK_OK) {
            delete *out;
            *out = nullptr;
        }
        return rc;
    });
}

int srack_patch_save_srk(const srack_patch* p, void* buf, size_t cap, size_t* n_bytes)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        // The app saves the running rack: every module struct as it is at that moment.  With srack_patch_keep_state (voices running on
        // across edits) the file therefore carries the CURRENT state — of voice 0, a rack file being one instance — not the stored one.
        // (Port buffers are written as stored: only the sink of a broken feedback edge ever reads them.)
        PatchHandle& h = const_cast<srack_patch*>(p)->h;
        std::vector<uint8_t> bytes;
        if (h.keep_state && h.samples_rendered > 0) {
            // the running state: on the device while the program that rendered is still there (also after an edit the next render
            // has not picked up yet); otherwise where the last re-flatten committed it (fields, per-voice overrides: voice 0).
            // A state field the host wrote since keeps the host's value.
            Graph snap = h.graph;
            std::vector<double> values;
            for (int m = 0; m < (int)snap.modules.size(); m++) {
                Module& mod = snap.modules[(size_t)m];
                bool ran = false;
                for (int f = 0; f < (int)mod.fields.size(); f++) {
                    if (!Graph::field_is_state(mod.type, f) || h.state_writes.count({m, f})) continue;
                    if (h.prog_valid && h.dev && read_device_state(h, m, f, values)) {
                        mod.fields[(size_t)f] = values[0];
                        ran = true;
                    } else {
                        for (const auto& o : h.overrides)
                            if (o.module == m && o.field == f && !o.values.empty()) mod.fields[(size_t)f] = o.values[0];
                    }
                }
                if (ran && mod.type == SRACK_MOD_SAMPLE && mod.wave_revision <= h.prog_graph_revision) mod.fields[SRACK_SAMPLE_WAVE_NEW] = 0.0;  // consumed by the first tick (sample.rs:199-203)
            }
            bytes = save_srk(snap);
        } else {
            bytes = save_srk(h.graph);
        }
        if (n_bytes) *n_bytes = bytes.size();
        if (buf && cap < bytes.size()) {  // never hand back a truncated file that still parses as MessagePack up to the cut
            set_error("save_srk: buffer too small (" + std::to_string(cap) + " bytes, the file needs " + std::to_string(bytes.size()) + "); call with buf = NULL for the size");
            return SRACK_ERR_INVALID;
        }
        if (buf) std::memcpy(buf, bytes.data(), bytes.size());
        return SRACK_OK;
    });
}

int srack_patch_module_id(const srack_patch* p, int module, char* buf, size_t cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const Module* m = get_module(p, module);
        if (!m) return SRACK_ERR_INVALID;
        if (buf && cap) std::snprintf(buf, cap, "%s", m->id.c_str());
        return (int)m->id.size();
    });
}

int srack_patch_set_module_position(srack_patch* p, int module, float x, float y)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (!get_module(p, module)) return SRACK_ERR_INVALID;
        Module& m = p->h.graph.modules[(size_t)module];
        m.has_pos = true;
        m.pos_x = x;
        m.pos_y = y;
        return SRACK_OK;
    });
}

int srack_patch_get_module_position(const srack_patch* p, int module, float* x, float* y)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const Module* m = get_module(p, module);
        if (!m) return SRACK_ERR_INVALID;
        if (x) *x = m->pos_x;
        if (y) *y = m->pos_y;
        return m->has_pos ? 1 : 0;
    });
}

int srack_patch_set_output_buffer(srack_patch* p, int module, int port, const float* samples, uint32_t n)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.set_output_buffer(module, port, samples, n);
    });
}

int srack_patch_get_output_buffer(const srack_patch* p, int module, int port, float* dst, uint32_t cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const Graph& g = p->h.graph;
        if (module < 0 || module >= (int)g.modules.size()) {
            set_error("get_output_buffer: no such module");
            return SRACK_ERR_INVALID;
        }
        const Module& m = g.modules[(size_t)module];
        if (port < 0 || port >= m.n_out) {
            set_error("get_output_buffer: no such port");  // Err(()) of get_output
            return SRACK_ERR_PORT;
        }
        if ((size_t)port >= m.out_init.size()) return 0;
        const std::vector<float>& b = m.out_init[(size_t)port];
        if (dst)
            for (size_t i = 0; i < b.size() && i < (size_t)cap; i++) dst[i] = b[i];
        return (int)b.size();
    });
}

int srack_patch_keep_state(srack_patch* p, int keep)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (p->h.keep_state != (keep != 0)) p->h.graph.revision++;  // (the flattened program differs: with keep, every planned module is evaluated)
        p->h.keep_state = keep != 0;
        return SRACK_OK;
    });
}

int srack_patch_set_noise_seed(srack_patch* p, uint64_t seed, uint64_t first_voice)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        p->h.graph.cfg.noise_seed = seed;
        p->h.graph.cfg.noise_first_voice = first_voice;
        p->h.graph.revision++;  // the keys are part of the flattened program
        return SRACK_OK;
    });
}

int srack_patch_connect(srack_patch* p, int src_module, int src_port, int sink_module, int sink_port)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.connect(src_module, src_port, sink_module, sink_port);
    });
}

int srack_patch_disconnect(srack_patch* p, int sink_module, int sink_port)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return p->h.graph.disconnect(sink_module, sink_port);
    });
}

int srack_patch_get_input(const srack_patch* p, int sink_module, int sink_port, int* src_module, int* src_port)
{
    return guarded([&]() -> int {
        const Module* m = get_module(p, sink_module);
        if (!m) return SRACK_ERR_INVALID;
        if (sink_port < 0 || sink_port >= m->n_in) {
            set_error("get_input: port index out of range");
            return SRACK_ERR_PORT;
        }
        if (src_module) *src_module = m->in[(size_t)sink_port].src;
        if (src_port) *src_port = m->in[(size_t)sink_port].port;
        return SRACK_OK;
    });
}

int srack_patch_plan(srack_patch* p, int* order, int cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        Graph& g = p->h.graph;
        int n = g.make_plan();
        if (g.plan.output < 0) {
            set_error("plan: no OutputModule in the module list (plan is empty)");
            return SRACK_ERR_NO_OUTPUT;
        }
        for (int i = 0; i < n && i < cap && order; i++) order[i] = g.plan.order[(size_t)i];
        return n;
    });
}

int srack_patch_plan_list(srack_patch* p, int output, const int* all_modules, int n_all, int* order, int cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        Graph& g = p->h.graph;
        const int n_mod = (int)g.modules.size();
        if (output < 0 || output >= n_mod || !all_modules || n_all < 0) {
            set_error("plan_list: bad output / list");
            return SRACK_ERR_INVALID;
        }
        std::vector<int> all(all_modules, all_modules + n_all);
        for (int m : all)
            if (m < 0 || m >= n_mod) {
                set_error("plan_list: module index out of range");
                return SRACK_ERR_INVALID;
            }
        int n = g.make_plan(output, all);
        for (int i = 0; i < n && i < cap && order; i++) order[i] = g.plan.order[(size_t)i];
        g.plan.valid = false;  // a shuffled list is a test device; renders always plan in list order
        return n;
    });
}

int srack_patch_removed_edges(srack_patch* p, int* pairs, int cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        const auto& r = p->h.graph.plan.removed;
        for (size_t i = 0; i < r.size() && (int)i < cap && pairs; i++) {
            pairs[2 * i] = r[i].first;
            pairs[2 * i + 1] = r[i].second;
        }
        return (int)r.size();
    });
}

int srack_patch_delayed_edges(srack_patch* p, int* quads, int cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        Graph& g = p->h.graph;
        if (!g.plan.valid) g.make_plan();
        auto edges = g.delayed_edges();
        for (size_t i = 0; i < edges.size() && (int)i < cap && quads; i++) {
            quads[4 * i + 0] = edges[i].src;
            quads[4 * i + 1] = edges[i].src_port;
            quads[4 * i + 2] = edges[i].sink;
            quads[4 * i + 3] = edges[i].sink_port;
        }
        return (int)edges.size();
    });
}

int srack_voices_configure(srack_patch* p, uint32_t n_voices)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (n_voices == 0 || n_voices > (1u << 24)) {  // a tile of 32 frame rows (32 * V * 4 bytes) must fit a 31-bit buffer offset
            set_error("voices_configure: n_voices must be 1 .. 16777216");
            return SRACK_ERR_INVALID;
        }
        p->h.n_voices = n_voices;
        p->h.overrides.clear();
        p->h.voices_revision++;
        p->h.voices_fresh = true;
        return SRACK_OK;
    });
}

int srack_voices_set_field_f32(srack_patch* p, int module, int field, const float* values) { return set_voice_field(p, module, field, values); }
int srack_voices_set_field_f64(srack_patch* p, int module, int field, const double* values) { return set_voice_field(p, module, field, values); }

int srack_render_planes(srack_patch* p, int* channel_plane, int cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        int rc = ensure_program(p->h, p->h.prog_valid ? p->h.prog_flags : 0u);
        if (rc != SRACK_OK) return rc;
        const DevProgram& H = p->h.prog.voice.hdr;
        for (int c = 0; c < H.n_channels && c < cap && channel_plane; c++) channel_plane[c] = H.channel_plane[c];
        return H.n_planes;
    });
}

int srack_render(srack_patch* p, uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags, void* stream)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (p->h.n_voices == 0) {
            set_error("render: call srack_voices_configure first");
            return SRACK_ERR_STATE;
        }
        return device_render(p->h, n_samples, d_frames, d_mix, flags, stream);
    });
}

int srack_render_reserve(srack_patch* p, uint32_t n_samples, int want_mix, uint32_t flags)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (p->h.n_voices == 0) {
            set_error("render_reserve: call srack_voices_configure first");
            return SRACK_ERR_STATE;
        }
        return device_reserve(p->h, n_samples, want_mix != 0, flags);
    });
}

// Every SRACK_* variable of the environment that the library's tuning knobs read (tools/: chunk lengths, generator experiments ...): a host
// process that carries one renders with other kernels than the ones measured and tested — srack_render_info says so (" knobs=[...]"),
// and the parity suites refuse to run with any set (tests/conftest.py).  Not listed: where the kernel cache lives and how much it keeps.
extern char** environ;
static std::string tuning_knobs_note()
{
    std::string s;
    for (char** e = environ; e && *e; e++) {
        if (std::strncmp(*e, "SRACK_", 6) != 0) continue;
        if (std::strncmp(*e, "SRACK_KERNEL_CACHE_", 19) == 0 || std::strncmp(*e, "SRACK_BENCH_", 12) == 0 || std::strncmp(*e, "SRACK_TEST_", 11) == 0) continue;
        s += s.empty() ? " knobs=[" : " ";
        s += *e;
    }
    if (!s.empty()) s += "]";
    return s;
}

int srack_render_info(srack_patch* p, char* buf, size_t cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        int rc = ensure_program(p->h, p->h.prog_valid ? p->h.prog_flags : 0u);
        if (rc != SRACK_OK) return rc;
        std::string s = p->h.prog.description;
        s += tuning_knobs_note();
        const char* k = device_kernel_name(p->h);
        // (the kernel's name stays LAST: hosts and tests read it with split("kernel="))
        s += device_jit_note(p->h);
        if (k && *k) s += std::string(" kernel=") + k;
        if (buf && cap) {
            std::strncpy(buf, s.c_str(), cap - 1);
            buf[cap - 1] = 0;
        }
        return (int)s.size();
    });
}

int srack_render_kernel_source(srack_patch* p, uint32_t flags, char* buf, size_t cap)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (p->h.n_voices == 0) {
            set_error("render_kernel_source: call srack_voices_configure first");
            return SRACK_ERR_STATE;
        }
        FlatPair scratch;
        const FlatPair* prog = nullptr;
        int rc = peek_program(p->h, flags, scratch, &prog);
        if (rc != SRACK_OK) return rc;
        std::string src;
        rc = jit_source(*prog, 3, jit_ctl_supported(*prog), src);
        if (rc != SRACK_OK) return rc;
        src = "// for: " + prog->description + "\n" + src;  // (for the reader; not part of what is compiled and cached)
        if (buf && cap) {
            std::strncpy(buf, src.c_str(), cap - 1);
            buf[cap - 1] = 0;
        }
        return (int)src.size();
    });
}

int srack_render_kernel_compile(srack_patch* p, uint32_t flags)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        if (p->h.n_voices == 0) {
            set_error("render_kernel_compile: call srack_voices_configure first");
            return SRACK_ERR_STATE;
        }
        FlatPair scratch;
        const FlatPair* prog = nullptr;
        int rc = peek_program(p->h, flags, scratch, &prog);
        if (rc != SRACK_OK) return rc;
        return jit_compile_only(*prog, 3, jit_ctl_supported(*prog));
    });
}

int srack_render_kernel_ms(srack_patch* p, double* avg_ms, int* n_launches, int reset)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        return device_kernel_ms(p->h, avg_ms, n_launches, reset);
    });
}

int srack_voices_get_field(srack_patch* p, int module, int field, double* values)
{
    return guarded([&]() -> int {
        CHECK_HANDLE(p);
        PatchHandle& h = p->h;
        if (!values || h.n_voices == 0) {
            set_error("voices_get_field: bad arguments / voices not configured");
            return SRACK_ERR_INVALID;
        }
        int rc = ensure_program(h, h.prog_valid ? h.prog_flags : 0u);
        if (rc != SRACK_OK) return rc;
        std::vector<double> got;
        if (read_device_state(h, module, field, got)) {
            for (uint32_t v = 0; v < h.n_voices; v++) values[v] = got[v];
            return SRACK_OK;
        }
        // a parameter, or a module that is not evaluated: the field value itself
        double x;
        rc = h.graph.get_field(module, field, &x);
        if (rc != SRACK_OK) return rc;
        for (uint32_t v = 0; v < h.n_voices; v++) values[v] = x;
        for (const auto& o : h.overrides)
            if (o.module == module && o.field == field)
                for (uint32_t v = 0; v < h.n_voices; v++) values[v] = o.values[v];
        return SRACK_OK;
    });
}

extern "C++" {
namespace srack {
bool read_device_state(PatchHandle& h, int module, int field, std::vector<double>& values)
{
    if (!h.prog_valid || !h.dev || module < 0 || module >= (int)h.graph.modules.size()) return false;
    // a module evaluated by the control program has ONE state, shared by every voice
    const int stage = h.prog.n_tracks > 0 && module < (int)h.prog.ctl_stage.size() ? h.prog.ctl_stage[(size_t)module] : -1;
    const bool ctl = stage >= 0;
    const FlatProgram& P = ctl ? h.prog.ctl[(size_t)stage] : h.prog.voice;
    const StateLoc loc = P.locate(h.graph, module, field);
    if (loc.row < 0) return false;
    const uint32_t V = P.n_voices;
    std::vector<uint32_t> rows((size_t)V * (loc.f64 ? 2 : 1));
    if (device_read_rows(h, stage, loc.row, loc.f64 ? 2 : 1, rows.data()) != SRACK_OK) return false;
    values.resize(h.n_voices);
    for (uint32_t v = 0; v < h.n_voices; v++) {
        const uint32_t sv = ctl ? 0u : v;
        if (loc.f64) {
            uint64_t u = (uint64_t)rows[sv] | ((uint64_t)rows[(size_t)V + sv] << 32);
            if (loc.fixed64)
                values[v] = std::ldexp((double)u, -64);  // phase * 2^64 (the fused kernel's fixed-point oscillator)
            else
                std::memcpy(&values[v], &u, 8);
        } else if (loc.flag) {
            values[v] = (double)(int32_t)rows[sv];
        } else {
            float f;
            std::memcpy(&f, &rows[sv], 4);
            values[v] = (double)f;
        }
    }
    return true;
}
}  // namespace srack
}  // extern "C++"

// ---- the kernel cache (jit.cpp) ---------------------------------------------------------------------
int srack_kernel_cache_set_dir(const char* dir)
{
    return guarded([&]() -> int { return jit_cache_set_dir(dir); });
}

int srack_kernel_cache_stats(srack_kernel_cache_info* out)
{
    return guarded([&]() -> int {
        if (!out) {
            set_error("srack_kernel_cache_stats: out is null");
            return SRACK_ERR_INVALID;
        }
        const JitCacheStats st = jit_cache_stats();
        std::memset(out, 0, sizeof *out);
        out->compiled = st.compiled;
        out->disk_hits = st.disk_hits;
        out->memory_hits = st.memory_hits;
        out->modules_loaded = st.modules_loaded;
        out->code_evictions = st.code_evictions;
        out->module_evictions = st.module_evictions;
        out->resident_code_objects = st.resident_code_objects;
        out->resident_modules = st.resident_modules;
        out->compile_ms = st.compile_ms;
        std::snprintf(out->directory, sizeof out->directory, "%s", st.directory);
        return SRACK_OK;
    });
}

// ---- device helpers -------------------------------------------------------------------------------
int srack_device_count(int* n)
{
    return guarded([&]() -> int {
        int c = 0;
        hipError_t e = hipGetDeviceCount(&c);
        if (e != hipSuccess) c = 0;
        if (n) *n = c;
        return SRACK_OK;
    });
}

int srack_device_set(int device)
{
    return guarded([&]() -> int {
        HIP_TRY_C(hipSetDevice(device));
        return SRACK_OK;
    });
}

int srack_device_get(int* device, char* pci_bus_id, size_t cap)
{
    return guarded([&]() -> int {
        int dev = -1;
        HIP_TRY_C(hipGetDevice(&dev));
        if (device) *device = dev;
        if (pci_bus_id && cap > 0) {
            pci_bus_id[0] = 0;
            HIP_TRY_C(hipDeviceGetPCIBusId(pci_bus_id, (int)std::min<size_t>(cap, 1u << 20), dev));
        }
        return SRACK_OK;
    });
}

int srack_device_alloc(void** d_ptr, size_t bytes)
{
    return guarded([&]() -> int {
        if (!d_ptr) return SRACK_ERR_INVALID;
        HIP_TRY_C(hipMalloc(d_ptr, bytes));
#if defined(SRK_POISON) && SRK_POISON
        HIP_TRY_C(hipMemset(*d_ptr, 0xff, bytes));
#endif
        return SRACK_OK;
    });
}

int srack_device_free(void* d_ptr)
{
    return guarded([&]() -> int {
        HIP_TRY_C(hipFree(d_ptr));
        return SRACK_OK;
    });
}

// Device -> host through PINNED bounce buffers of the library's own, a chunk at a time (DMA into pinned memory, then a CPU copy).  A plain
// hipMemcpyAsync into the caller's pageable memory FOLLOWED BY hipStreamSynchronize on the same stream is what this was until round 5 — and
// what that round's soaks caught losing data: with sixteen processes on one device, one read-back in a few thousand came back with a stretch
// of ZEROS (tens of pages, the same offsets in call after call of one process) where a second read-back of the very same device bytes had the
// data (tools/fuzz_soak_default.py, SOAK_RETRY=2; notes/r05.md R5.2).  It was not a copy the host read too early — the synchronize was there,
// on the copy's own stream, before the call returned (git show 0c84165^:s-rack_amd/csrc/capi.cpp) —: the runtime's staged copy into pageable
// pages itself dropped them.  The caller's pages are never handed to the copy engine now.
//   * per DEVICE a small pool of bounce sets (two pinned halves of kBounceBytes, a non-blocking stream, two events), created on demand and
//     freed when the library is unloaded; no lock is held across a copy (a set is taken out of the pool for the duration);
//   * the copy of chunk k + 1 runs (DMA) under the CPU's memcpy of chunk k;
//   * a large read-back is cut into slices that helper threads copy side by side, each through a set of its own: one core's memcpy into
//     fresh pageable memory (page faults included) is a few GB/s, PCIe is tens.
// Ordering: everything enqueued on `stream` before the call is waited for first (hipStreamSynchronize), as before.
namespace {
constexpr size_t kBounceBytes = size_t(8) << 20;
constexpr size_t kSliceMin = size_t(64) << 20;   // a helper thread is worth it from here
constexpr int kMaxHelpers = 8;
struct BounceSet {
    int device = -1;
    void* half[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    hipStream_t st = nullptr;
};
struct BouncePool {
    std::mutex m;
    std::vector<BounceSet*> idle;
    ~BouncePool()
    {   // library unload / process exit: the HIP runtime may already be gone — errors are ignored
        for (BounceSet* b : idle) drop(b);
    }
    int take(int device, BounceSet** out)
    {
        {
            std::lock_guard<std::mutex> lock(m);
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i]->device == device) {
                    *out = idle[i];
                    idle.erase(idle.begin() + (long)i);
                    return SRACK_OK;
                }
        }
        BounceSet* b = new BounceSet();
        b->device = device;
        auto build = [&]() -> int {
            for (int k = 0; k < 2; k++) {
                HIP_TRY_C(hipHostMalloc(&b->half[k], kBounceBytes, hipHostMallocDefault));
                HIP_TRY_C(hipEventCreateWithFlags(&b->done[k], hipEventDisableTiming));
            }
            HIP_TRY_C(hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking));
            return SRACK_OK;
        };
        const int rc = build();
        if (rc != SRACK_OK) {  // a half-built set never reaches the pool (the error text stays the failing call's)
            drop(b);
            return rc;
        }
        *out = b;
        return SRACK_OK;
    }
    static void drop(BounceSet* b)
    {
        for (int k = 0; k < 2; k++) {
            if (b->half[k]) (void)hipHostFree(b->half[k]);
            if (b->done[k]) (void)hipEventDestroy(b->done[k]);
        }
        if (b->st) (void)hipStreamDestroy(b->st);
        delete b;
    }
    void give(BounceSet* b)
    {
        if (!b) return;
        std::lock_guard<std::mutex> lock(m);
        idle.push_back(b);
    }
};
BouncePool g_bounce_pool;

// one slice, through one set: DMA of chunk k + 1 under the memcpy of chunk k
int bounce_copy(int device, char* dst, const char* src, size_t bytes)
{
    HIP_TRY_C(hipSetDevice(device));  // (helper threads start on device 0)
    BounceSet* b = nullptr;
    int rc = g_bounce_pool.take(device, &b);
    auto run = [&]() -> int {
        if (rc != SRACK_OK) return rc;
        const size_t n_chunks = (bytes + kBounceBytes - 1) / kBounceBytes;
        auto len = [&](size_t k) { return std::min(kBounceBytes, bytes - k * kBounceBytes); };
        auto issue = [&](size_t k) -> int {
            HIP_TRY_C(hipMemcpyAsync(b->half[k & 1], src + k * kBounceBytes, len(k), hipMemcpyDeviceToHost, b->st));
            HIP_TRY_C(hipEventRecord(b->done[k & 1], b->st));
            return SRACK_OK;
        };
        int r = issue(0);
        for (size_t k = 0; k < n_chunks && r == SRACK_OK; k++) {
            if (k + 1 < n_chunks) r = issue(k + 1);  // (its half was emptied by the memcpy of chunk k - 1)
            if (r != SRACK_OK) break;
            HIP_TRY_C(hipEventSynchronize(b->done[k & 1]));
            std::memcpy(dst + k * kBounceBytes, b->half[k & 1], len(k));
        }
        if (r != SRACK_OK) (void)hipStreamSynchronize(b->st);  // nothing of ours in flight when the set goes back
        return r;
    };
    rc = run();
    g_bounce_pool.give(b);
    return rc;
}
}  // namespace
int srack_device_to_host(void* h_dst, const void* d_src, size_t bytes, void* stream)
{
    return guarded([&]() -> int {
        if (bytes == 0) return SRACK_OK;
        if (!h_dst || !d_src) return SRACK_ERR_INVALID;
        int device = 0;
        HIP_TRY_C(hipGetDevice(&device));
        HIP_TRY_C(hipStreamSynchronize((hipStream_t)stream));  // what the caller enqueued before the copy
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int helpers = (int)std::min<size_t>({(size_t)kMaxHelpers, (size_t)hw, bytes / kSliceMin});
        if (helpers <= 1) return bounce_copy(device, (char*)h_dst, (const char*)d_src, bytes);
        // slices of whole chunks, one helper thread each (the calling thread takes the first)
        const size_t n_chunks = (bytes + kBounceBytes - 1) / kBounceBytes, per = (n_chunks + (size_t)helpers - 1) / (size_t)helpers * kBounceBytes;
        std::vector<int> rcs((size_t)helpers, SRACK_OK);
        std::vector<std::string> errs((size_t)helpers);
        std::vector<std::thread> threads;
        auto slice = [&](int i) {
            const size_t off = (size_t)i * per;
            if (off >= bytes) return;
            rcs[(size_t)i] = guarded([&]() { return bounce_copy(device, (char*)h_dst + off, (const char*)d_src + off, std::min(per, bytes - off)); });
            if (rcs[(size_t)i] != SRACK_OK) errs[(size_t)i] = srack_last_error();  // (thread-local: carried back to the caller's thread below)
        };
        int started = 1;  // (slice 0 is the caller's)
        try {
            for (; started < helpers; started++) threads.emplace_back(slice, started);
        } catch (const std::exception&) {  // no more threads to be had: the slices without one are copied here, after the caller's own
        }
        slice(0);
        for (int i = started; i < helpers; i++) slice(i);
        for (std::thread& t : threads) t.join();
        for (int i = 0; i < helpers; i++)
            if (rcs[(size_t)i] != SRACK_OK) {
                if (i > 0) set_error(errs[(size_t)i]);
                return rcs[(size_t)i];
            }
        return SRACK_OK;
    });
}

int srack_device_sync(void* stream)
{
    return guarded([&]() -> int {
        HIP_TRY_C(hipStreamSynchronize((hipStream_t)stream));
        return SRACK_OK;
    });
}

}  // extern "C"
