// backward_emu.cpp -- rlipv2_amd/csrc/msda_patch.hip ITSELF (kernels and their launchers) compiled for the CPU against the
// lane-level workgroup model of tools/emu/stub/hip/hip_runtime.h: the encoder backward's cell + patch route
// (cell_backward_kernel -> patch_dest_kernel) on a small problem.  Test infrastructure (tests/test_backward_emulated.py).
// The ablation switches of the launchers are live (MSDA_ABLATION): RLIPV2_CELL_SHARED, RLIPV2_PATCH_MULTI, RLIPV2_PATCH_REPS,
// RLIPV2_PATCH_CELLG
// select the experiment arms, so their LOGIC can be checked against the oracle -- and against the product kernels, which a
// GPU has validated and which therefore calibrate the model -- without a GPU.
// usage: backward_emu problem.bin out.bin     (problem: see main; out: g_value bf16 | g_loc f32 | g_aw f32 | far flag int32;
//        EMU_FUSED=1: g_value bf16 | gradient of the projection rows bf16 [N * Lq, M * 48] | far flag)
#define MSDA_EMU 1
#define MSDA_ABLATION 1
#include <hip/hip_runtime.h>

#include "../../rlipv2_amd/csrc/msda_patch.hip"

namespace msda {
bool quad_supports(const Problem &) { return true; }       // (msda_quad.hip: addressing limits, irrelevant at this size)
}

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s problem.bin out.bin\n", argv[0]); return 2; }
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t h[12];
    if (std::fread(h, 4, 12, f) != 12) return 2;
    const int N = h[0], S = h[1], M = h[2], Lq = h[3];
    int64_t hs[8], shapes[8], starts[4];
    for (int k = 0; k < 8; ++k) hs[k] = shapes[k] = h[4 + k];
    std::vector<uint16_t> value((size_t)N * S * M * 32), go((size_t)N * Lq * M * 32), gv((size_t)N * S * M * 32, 0x7fc0);
    std::vector<float> loc((size_t)N * Lq * M * 32), aw((size_t)N * Lq * M * 16), gl(loc.size(), NAN), ga(aw.size(), NAN);
    if (std::fread(value.data(), 2, value.size(), f) != value.size()) return 2;
    if (std::fread(starts, 8, 4, f) != 4) return 2;
    if (std::fread(loc.data(), 4, loc.size(), f) != loc.size()) return 2;
    if (std::fread(aw.data(), 4, aw.size(), f) != aw.size()) return 2;
    if (std::fread(go.data(), 2, go.size(), f) != go.size()) return 2;
    std::fclose(f);
    msda::Problem p{};
    p.dtype = MSDA_BF16; p.N = N; p.S = S; p.M = M; p.D = 32; p.L = 4; p.Lq = Lq; p.P = 4;
    p.value = value.data(); p.shapes = shapes; p.starts = starts; p.loc = loc.data(); p.aw = aw.data();
    p.grad_out = go.data(); p.g_value = gv.data(); p.g_loc = gl.data(); p.g_aw = ga.data(); p.stream = nullptr;
    if (!msda::cell_backward_supports(p, hs)) { std::fprintf(stderr, "emu: the cell + patch route does not take this problem\n"); return 3; }
    const size_t wsb = msda::patch_workspace_bytes(p, hs);
    std::vector<unsigned char> ws(wsb + 64, 0xa5);                 // (workspace contents are garbage on entry)
    alignas(16) int ctl[64] = {0};
    // EMU_FUSED=1: the route of the train step -- the module's geometry backward as the kernel's epilogue (reference points
    // = the pixel centres of the encoder, refdim 2); writes the gradient of the projection row instead of g_loc / g_aw
    const bool fused = std::getenv("EMU_FUSED") && std::atoi(std::getenv("EMU_FUSED")) != 0;
    std::vector<float> ref((size_t)N * Lq * 4 * 2);
    std::vector<uint16_t> gq((size_t)N * Lq * M * 48, 0x7fc0);
    msda::Fused fz{};
    if (fused) {
        for (int n = 0; n < N; ++n) {
            int q = 0;
            for (int lq = 0; lq < 4; ++lq)
                for (int y = 0; y < hs[2 * lq]; ++y)
                    for (int x = 0; x < hs[2 * lq + 1]; ++x, ++q)
                        for (int l = 0; l < 4; ++l) {
                            ref[(((size_t)n * Lq + q) * 4 + l) * 2] = ((float)x + 0.5f) / (float)hs[2 * lq + 1];
                            ref[(((size_t)n * Lq + q) * 4 + l) * 2 + 1] = ((float)y + 0.5f) / (float)hs[2 * lq];
                        }
        }
        fz.ref = ref.data(); fz.refdim = 2; fz.g_qproj = gq.data();
    }
    msda::launch_cell_backward(p, fused ? &fz : nullptr, hs, ctl, ws.data());
    const size_t gco = msda::patch_gcell_offset(p, hs);                   // RLIPV2_PATCH_CELLG=1: the cell-major grad_out copy
    msda::launch_patch_dest(p, hs, ctl, ws.data(), true, true, gco ? ws.data() + gco : nullptr);
    f = std::fopen(argv[2], "wb");
    std::fwrite(gv.data(), 2, gv.size(), f);
    if (fused) std::fwrite(gq.data(), 2, gq.size(), f);
    else {
        std::fwrite(gl.data(), 4, gl.size(), f);
        std::fwrite(ga.data(), 4, ga.size(), f);
    }
    const int32_t far = ctl[60];
    std::fwrite(&far, 4, 1, f);
    std::fclose(f);
    return 0;
}
