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

// the table form of the same lines (wf_plan / wf_op, what k_wire.hip expands lines with): every emit call of the walk, those
// outside the surface included, must be the table's entry of the same index.  Returns the first line that differs, or -1.
namespace {
struct Collect {
    int *xya; int n, cap;
    void operator()(int x, int y, uint32_t alpha) { if (n < cap) { xya[3 * n] = x; xya[3 * n + 1] = y; xya[3 * n + 2] = (int)alpha; } n++; }
};
}

extern "C" int emu_wire_check_table(int W, int H, int n, const int16_t *xyxy)
{
    static int buf[3 * 20000];
    for (int l = 0; l < n; l++) {
        Collect c{buf, 0, 20000};
        wf_aaline(W, H, xyxy[4 * l], xyxy[4 * l + 1], xyxy[4 * l + 2], xyxy[4 * l + 3], c);
        const WfPlan p = wf_plan(W, H, xyxy[4 * l], xyxy[4 * l + 1], xyxy[4 * l + 2], xyxy[4 * l + 3]);
        if ((int)p.n != c.n || c.n > c.cap) return l;
        for (uint32_t i = 0; i < p.n; i++) {
            int x, y; uint32_t a;
            wf_op(p, i, x, y, a);
            if (x != buf[3 * i] || y != buf[3 * i + 1] || (int)a != buf[3 * i + 2]) return l;
        }
    }
    return -1;
}

// ... and drawn through the table (what the sorted replay of k_wire.hip amounts to for lines drawn one after the other)
extern "C" void emu_wire_lines_table(uint32_t *pixels, int W, int H, int pitch_words, int n, const int16_t *xyxy)
{
    Direct d{pixels, pitch_words, W, H};
    for (int l = 0; l < n; l++) {
        const WfPlan p = wf_plan(W, H, xyxy[4 * l], xyxy[4 * l + 1], xyxy[4 * l + 2], xyxy[4 * l + 3]);
        for (uint32_t i = 0; i < p.n; i++) { int x, y; uint32_t a; wf_op(p, i, x, y, a); d(x, y, a); }
    }
}
