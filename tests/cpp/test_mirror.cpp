// The reference's own two tests, written against the C++ mirror of its API (include/srack.hpp).
//   test_mirror topo   — synth::tests::topological_sort (src/synth.rs:537-613), CPU only
//   test_mirror dco    — oscillator::dco_tests::produces_440 (src/synth/oscillator.rs:284-305), needs the GPU
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../../include/srack.hpp"

using namespace srack;

#define REQUIRE(cond)                                                      \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);  \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static void connect(const SharedSynthModule& src, SharedSynthModule sink)
{
    auto inputs = get_inputs(sink);  // first unconnected input (synth.rs:523-535)
    size_t idx = 0;
    while (inputs[idx].has_value()) idx++;
    if (!sink.set_input((uint8_t)idx, src, 0)) throw Error(SRACK_ERR_PORT, "set_input");
}

static int topological_sort()
{
    //     0 -> 1 -> 2 -> 3 -> o
    //      \----> 4 -----^
    //        5<->6^
    AudioConfig ac;
    ac.buffer_size = 64;
    ac.sample_rate = 44100;
    ac.channels = 2;
    Workspace ws(ac);
    std::vector<SharedSynthModule> modules;
    for (int i = 0; i < 7; i++) modules.push_back(ws.add(ModuleType::MonoMixer));
    SharedSynthModule out = ws.add(ModuleType::Output);
    connect(modules[0], modules[1]);
    connect(modules[1], modules[2]);
    connect(modules[2], modules[3]);
    connect(modules[3], out);
    connect(modules[0], modules[4]);
    connect(modules[4], modules[3]);
    connect(modules[6], modules[4]);
    connect(modules[5], modules[6]);
    connect(modules[6], modules[5]);
    std::mt19937 rng(12345);
    for (int iter = 0; iter < 1000; iter++) {
        std::vector<SharedSynthModule> list(modules), plan;
        list.push_back(out);
        std::shuffle(list.begin(), list.end(), rng);
        plan_execution(ws, out, list, plan);
        std::map<int, size_t> indexes;
        for (size_t i = 0; i < plan.size(); i++) indexes[plan[i].index()] = i;
        auto at = [&](const SharedSynthModule& m) { return indexes.at(m.index()); };
        REQUIRE(plan.size() == 8);
        REQUIRE(at(modules[0]) < at(modules[1]));
        REQUIRE(at(modules[1]) < at(modules[2]));
        REQUIRE(at(modules[2]) < at(modules[3]));
        REQUIRE(at(modules[3]) < at(out));
        REQUIRE(at(modules[0]) < at(modules[4]));
        REQUIRE(at(modules[4]) < at(modules[3]));
        REQUIRE(at(modules[6]) < at(modules[4]));
        REQUIRE(at(modules[5]) < at(modules[6]));
    }
    // error behaviour of the trait: Err(()) on a bad port, None for an unconnected input
    REQUIRE(!modules[0].set_input(9, modules[1], 0));
    REQUIRE(!modules[0].get_input(9).has_value());
    REQUIRE(modules[2].get_input(3).has_value() && !modules[2].get_input(3)->has_value());
    REQUIRE(modules[1].get_input(0)->value().first == modules[0]);
    REQUIRE(out.get_num_inputs() == 2 && out.get_num_outputs() == 0 && modules[0].get_name() == "Mono Mixer");
    {   // every module type of the reference has a mirror, named as its get_name() (synth.rs:421-470)
        Workspace all(ac);
        const char* names[] = {"Output", "Oscillator", "Moog Filter", "ADSR", "VCA", "Mono Mixer", "Add", "Grid Sequencer", "Pattern Sequencer",
                               "Non-Linear", "Sample", "Noise", "Freeverb"};
        for (int t = 0; t < SRACK_MOD__COUNT; t++) REQUIRE(all.add((ModuleType)t).get_name() == names[t]);
        SharedSynthModule nz = all.add(ModuleType::Noise), fv = all.add(ModuleType::Freeverb);
        REQUIRE(nz.get_num_inputs() == 0 && nz.get_num_outputs() == 1 && fv.get_num_inputs() == 2 && fv.get_num_outputs() == 2);
        REQUIRE(fv.set_input(1, nz, 0) && !nz.set_input(0, fv, 0));
    }
    std::printf("topological_sort ok\n");
    return 0;
}

static int produces_440()
{
    AudioConfig ac;
    ac.sample_rate = 440 * 4;
    ac.buffer_size = 17;  // notice odd sized buffer
    ac.channels = 2;
    Workspace ws(ac);
    SharedSynthModule module = ws.add(ModuleType::Oscillator);
    SharedSynthModule out = ws.add(ModuleType::Output);
    REQUIRE(out.set_input(0, module, SRACK_OSC_OUT_SINE));
    ws.configure_voices(1);
    REQUIRE(ws.planes() == 1);
    void* d = nullptr;
    REQUIRE(srack_device_alloc(&d, 17 * sizeof(float)) == SRACK_OK);
    float buf[17];
    ws.execute_batch(17, (float*)d, nullptr);  // module.calc()
    REQUIRE(srack_device_to_host(buf, d, sizeof(buf), nullptr) == SRACK_OK);
    REQUIRE(buf[0] == 0.0f);
    REQUIRE(std::fabs(buf[1] - 1.0f) < 0.00001f);
    REQUIRE(std::fabs(buf[2]) < 0.00001f);
    REQUIRE(std::fabs(buf[3] + 1.0f) < 0.00001f);
    REQUIRE(std::fabs(buf[4]) < 0.00001f);
    ws.execute_batch(17, (float*)d, nullptr);  // module.calc() again
    REQUIRE(srack_device_to_host(buf, d, sizeof(buf), nullptr) == SRACK_OK);
    REQUIRE(std::fabs(buf[0] - 1.0f) < 0.00001f);  // should continue smoothly into next buffer
    srack_device_free(d);
    std::printf("produces_440 ok\n");
    return 0;
}

// The multi-GPU surface with one rank (a 1-GPU box): unique id -> communicator -> count -> the mix reduce through it.
static int one_rank_mix_comm()
{
    AudioConfig ac;
    ac.sample_rate = 48000;
    ac.buffer_size = 64;
    ac.channels = 2;
    Workspace ws(ac);
    SharedSynthModule osc = ws.add(ModuleType::Oscillator);
    SharedSynthModule out = ws.add(ModuleType::Output);
    REQUIRE(out.set_input(0, osc, SRACK_OSC_OUT_SAW));
    REQUIRE(out.set_input(1, osc, SRACK_OSC_OUT_SAW));
    ws.configure_voices(100);
    const uint32_t T = 512;
    float *d_mix = nullptr, *d_mix2 = nullptr;
    ws.check(srack_device_alloc((void**)&d_mix, 2 * T * sizeof(float)));
    ws.check(srack_device_alloc((void**)&d_mix2, 2 * T * sizeof(float)));
    ws.execute_batch(T, nullptr, d_mix);
    std::vector<float> before(2 * T), after(2 * T);
    ws.check(srack_device_to_host(before.data(), d_mix, before.size() * sizeof(float), nullptr));
    MixComm comm(MixComm::unique_id(), 1, 0);
    if (comm.count() != 1) return 1;
    comm.reduce_mix(d_mix, 2 * T, 0, nullptr);
    ws.check(srack_device_to_host(after.data(), d_mix, after.size() * sizeof(float), nullptr));
    for (size_t i = 0; i < before.size(); i++)
        if (before[i] != after[i]) return 1;  // a sum over one rank is the identity
    float peak = 0.0f;
    for (float v : before) peak = std::fmax(peak, std::fabs(v));
    if (!(peak > 10.0f)) return 1;  // 100 saws in phase
    srack_device_free(d_mix);
    srack_device_free(d_mix2);
    std::printf("one_rank_mix_comm ok\n");
    return 0;
}

// The audio callback of src/main.rs:59-90, transliterated: cpal hands over an interleaved `data: &mut [f32]` of whatever length it likes;
// whenever the staging buffers run dry (`src_buf_idx == 0` at a frame boundary) the callback runs `synth::execute(&plan)` — here one
// srack_render of buffer_size samples, the mix standing in for OutputModule::bufs — copies the channels out and goes on interleaving.
// A host that ticks like this must hear exactly what one long render gives (the library keeps its control program running ahead
// across such calls: a tick session).
static int audio_callback(uint32_t n_voices)
{
    AudioConfig ac;
    ac.sample_rate = 48000;
    ac.buffer_size = 256;
    ac.channels = 2;
    auto build = [&](Workspace& ws) {  // P1: saw VCO -> ladder -> VCA, envelope gated by an LFO square (a fast one: 110 Hz)
        SharedSynthModule osc = ws.add(ModuleType::Oscillator), lfo = ws.add(ModuleType::Oscillator), vcf = ws.add(ModuleType::MoogFilter);
        SharedSynthModule adsr = ws.add(ModuleType::ADSR), vca = ws.add(ModuleType::VCA), out = ws.add(ModuleType::Output);
        lfo.set(SRACK_OSC_VAL, -2.0);
        adsr.set(SRACK_ADSR_A_SEC, 0.001);
        adsr.set(SRACK_ADSR_D_SEC, 0.002);
        adsr.set(SRACK_ADSR_S_VAL, 0.5);
        adsr.set(SRACK_ADSR_R_SEC, 0.002);
        if (!vcf.set_input(0, osc, SRACK_OSC_OUT_SAW) || !adsr.set_input(0, lfo, SRACK_OSC_OUT_SQUARE) || !vca.set_input(0, vcf, 0) ||
            !vca.set_input(1, adsr, 0) || !out.set_input(0, vca, 0) || !out.set_input(1, vca, 0))
            throw Error(SRACK_ERR_PORT, "set_input");
        ws.configure_voices(n_voices);
        if (n_voices > 1) {
            std::vector<float> val(n_voices);
            for (uint32_t v = 0; v < n_voices; v++) val[v] = -0.5f + (float)v / (float)n_voices;
            ws.check(srack_voices_set_field_f32(ws.handle(), osc.index(), SRACK_OSC_VAL, val.data()));
        }
    };
    const size_t channels = ac.channels, buffer_size = ac.buffer_size;
    const size_t lens[] = {512, 2 * 256, 2 * 100, 2 * 733, 2 * 1, 2 * 1024, 2 * 256, 2 * 256, 2 * 77};  // data.len() per callback (frames x channels)
    size_t total = 0;
    for (size_t n : lens) total += n;
    // --- the ticking host ---
    Workspace ws(ac);
    build(ws);
    float* d_mix = nullptr;
    ws.check(srack_device_alloc((void**)&d_mix, channels * buffer_size * sizeof(float)));
    std::vector<std::vector<float>> src_buf(channels, std::vector<float>(buffer_size, 0.0f));
    std::vector<float> bufs(channels * buffer_size), heard;
    size_t src_buf_idx = 0, executes = 0;
    for (size_t n : lens) {
        std::vector<float> data(n);
        for (size_t out_idx = 0; out_idx < data.size(); out_idx++) {
            if (src_buf_idx == 0 && out_idx % channels == 0) {
                ws.execute_batch((uint32_t)buffer_size, nullptr, d_mix);  // synth::execute(&plan)
                executes++;
                ws.check(srack_device_to_host(bufs.data(), d_mix, bufs.size() * sizeof(float), nullptr));
                ws.check(srack_device_sync(nullptr));
                for (size_t c = 0; c < channels; c++) std::copy(bufs.begin() + c * buffer_size, bufs.begin() + (c + 1) * buffer_size, src_buf[c].begin());
            }
            data[out_idx] = src_buf[out_idx % channels][src_buf_idx];
            if (out_idx % channels == channels - 1) {
                src_buf_idx += 1;
                if (src_buf_idx >= buffer_size) src_buf_idx = 0;
            }
        }
        heard.insert(heard.end(), data.begin(), data.end());
    }
    srack_device_free(d_mix);
    // --- one long render of as many blocks ---
    Workspace whole(ac);
    build(whole);
    const size_t T = executes * buffer_size;
    float* d_all = nullptr;
    whole.check(srack_device_alloc((void**)&d_all, channels * T * sizeof(float)));
    whole.execute_batch((uint32_t)T, nullptr, d_all);
    std::vector<float> all(channels * T);
    whole.check(srack_device_to_host(all.data(), d_all, all.size() * sizeof(float), nullptr));
    whole.check(srack_device_sync(nullptr));
    srack_device_free(d_all);
    REQUIRE(heard.size() == total && total / channels <= T);
    float peak = 0.0f;
    for (size_t i = 0; i < heard.size(); i++) {
        const float want = all[(i % channels) * T + i / channels];
        REQUIRE(std::memcmp(&heard[i], &want, sizeof(float)) == 0);  // bit for bit
        peak = std::fmax(peak, std::fabs(want));
    }
    REQUIRE(peak > 0.05f * (float)(n_voices > 1 ? 8 : 1));
    std::printf("audio_callback ok (%zu executes, %zu frames, peak %.3f)\n", executes, total / channels, (double)peak);
    return 0;
}

int main(int argc, char** argv)
{
    try {
        if (argc > 1 && !std::strcmp(argv[1], "topo")) return topological_sort();
        if (argc > 1 && !std::strcmp(argv[1], "dco")) return produces_440();
        if (argc > 1 && !std::strcmp(argv[1], "dist")) return one_rank_mix_comm();
        if (argc > 1 && !std::strcmp(argv[1], "callback")) return audio_callback(argc > 2 ? (uint32_t)std::atoi(argv[2]) : 1u);
    } catch (const Error& e) {
        std::printf("srack::Error %d: %s\n", e.code, e.what());
        return 2;
    }
    std::printf("usage: test_mirror topo|dco|dist|callback [voices]\n");
    return 3;
}
