// host-only stand-in for render.hip: flatten on ensure_program, no device
#include "runtime.hpp"
namespace srack {
struct DeviceState {};
PatchHandle::~PatchHandle() {}
int ensure_program(PatchHandle& h, uint32_t flags)
{
    if (h.prog_valid && h.prog_graph_revision == h.graph.revision && h.prog_voices_revision == h.voices_revision && h.prog_flags == flags) return SRACK_OK;
    int rc = flatten(h.graph, h.n_voices, h.overrides, flags, h.prog);
    if (rc != SRACK_OK) return rc;
    h.prog_valid = true; h.prog_graph_revision = h.graph.revision; h.prog_voices_revision = h.voices_revision; h.prog_flags = flags;
    return SRACK_OK;
}
int device_render(PatchHandle&, uint32_t, float*, float*, uint32_t, void*) { set_error("no device"); return SRACK_ERR_DEVICE; }
int device_reserve(PatchHandle& h, uint32_t, bool, uint32_t flags) { return ensure_program(h, flags); }
int device_kernel_ms(PatchHandle&, double*, int*, int) { return SRACK_ERR_DEVICE; }
int device_read_rows(PatchHandle& h, int stage, int first_row, int n_rows, uint32_t* dst)
{
    const FlatProgram& P = stage >= 0 ? h.prog.ctl[(size_t)stage] : h.prog.voice;
    std::memcpy(dst, P.table.data() + (size_t)first_row * P.n_voices, sizeof(uint32_t) * P.n_voices * (size_t)n_rows);
    return SRACK_OK;
}
void device_release(DeviceState*) {}
const char* device_kernel_name(const PatchHandle&) { return "stub"; }
}
