// emu_wire.hip -- host build of the wireframe's line generator (renderer_amd/csrc/wf_core.h): the lines' pixel operations
// applied one after the other, as k_wire.hip's sorted replay applies them.  TEST INFRASTRUCTURE (tests/test_wire_emu.py).
#include "../../renderer_amd/csrc/wf_core.h"

#include <cstdint>

namespace {
struct Direct {
    uint32_t *px; int pitch, W, H;
    void operator()(int x, int y, uint32_t alpha)
    {
        if (!(x >= 0 && x < W && y >= 0 && y < H)) return;
        uint32_t &p = px[(size_t)y * pitch + x];
        p = wf_blend(p, alpha);
    }
};
}

extern "C" void emu_wire_lines(uint32_t *pixels, int W, int H, int pitch_words, int n, const int16_t *xyxy)
{
    Direct d{pixels, pitch_words, W, H};
    for (int i = 0; i < n; i++) wf_aaline(W, H, xyxy[4 * i], xyxy[4 * i + 1], xyxy[4 * i + 2], xyxy[4 * i + 3], d);
}
