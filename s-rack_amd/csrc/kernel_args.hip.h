// kernel_args.hip.h — the argument blocks shared by the kernels (interp.hip.h, fused.hip.h) and the launch code (render.hip).
#pragma once
#ifndef __HIPCC_RTC__
#include <cstdint>
#endif

#include "program.hpp"

namespace srack {

struct KernelArgs {
    const DevOp* ops;
    DevProgram prog;
    uint32_t* table;
    float* rings;
    float* frames;
    float* mixpart;
    const float* tracks;  // control tracks [n_tracks][t_stride] written by the control program (may be null)
    const uint32_t* seqtab;  // sequencer grids, 64 cells per sequencer op
    uint32_t V, T, n_waves;
    uint32_t lanes;  // voices per wave: 64, or 32 / 16 when there are too few voices to fill the SIMDs (idle lanes shadow the wave's last voice)
    // A launch covers T samples of a render of t_stride samples; frames / mixpart / tracks arrive pre-offset to
    // the launch's first sample and keep the whole render's strides.
    uint64_t plane_stride;  // frames: elements between planes (= t_stride * V)
    uint32_t t_stride;
    uint32_t block0;  // blocks [0, block0) of the grid are not voice waves (a co-scheduled control block); wave = blockIdx.x - block0
    uint64_t n0;  // absolute index of this launch's first sample (phase of the feedback rings)
    double* fv;   // OP_FREEVERB blocks: rows of V doubles (program.hpp), zero at n0 = 0 (may be null)
    // Kernels specialised at run time only (jit.cpp): blocks [0, block0) of a launch are the control program's units, block b
    // running unit b on the chunk ctl_slots[b] describes (T == 0: nothing to do in this launch).
    const KernelArgs* ctl_slots;
    // A control unit inside a tick session (render.hip, TickSession): the state this chunk leaves behind goes to another copy of the
    // table than the one it was read from, so that a chunk computed ahead of the host's next call never overwrites what the patch
    // holds as of the last rendered sample.  Null: in place.
    uint32_t* table_out;
};

struct ChainRoles {  // op indices of the fused voice chain (osc_l / adsr unused in the track variant)
    int osc_a, osc_l, vcf, adsr, vca, out, track;
};

struct SeqRoles {  // the fused sequencer-driven voice chain: op indices and rows of the track buffer
    int math, osc, vcf, vca, out;        // math = -1: the oscillator's CV is the note track itself
    int trk_pitch, trk_cutoff, trk_env;  // trk_cutoff = -1: the filter has no CV
    int n_extra;                         // further output planes that carry a track as it is
    int extra_plane[4], extra_trk[4];
};

}  // namespace srack
