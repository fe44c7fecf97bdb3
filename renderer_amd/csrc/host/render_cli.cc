// render_cli.cc -- headless counterpart of `renderer -b -n N -m MODE FILE` (renderer.cc:168-642).
//
// Reproduces the reference's benchmark loop -- same light and eye placement, same auto-spin
// camera sequence, BVH build and shadow-map generation outside the timed region, fps =
// frames / seconds spent inside Scene::render* -- on top of the C++ host API, i.e. on the GPU.
// There is no window: -o PREFIX dumps frames as binary PPM instead of SDL_Flip.
//   -r          report the frame rate every 5 seconds while running (renderer.cc:603-614)
//   --bench     the reference's `make bench` (src/Makefile.am:25-26): five runs of `-b -n 500` (or -n N), then
//               Average / Std dev / Median / Min / Max of their frame rates, as its perl one-liner prints them
//   -g N        draw every frame on N GPUs (devices 0..N-1; -g 0,0 lists devices explicitly)
//   -p N        keep N frames in flight (1..4, Scene::renderAsync): frame k is copied out while frame k+1 renders; the
//               reported rate is then frames / wall time of the loop (there is no "time inside render" to add up).
//               DEFAULT 3 on one device: in -b the cameras of the frames to come are known (renderer.cc:485-494: the orbit),
//               so nothing keeps the loop from asking for frame k + 1 before it shows frame k; -p 1 = the reference's loop,
//               one synchronous Scene::render* per pass, rate = frames / time inside those calls
//   --depth N   MAX_RAY_DEPTH (Raytracer.cc:56; default 3): 1 = primary + shadow rays, BASELINE.json's configs[2]
//   --keys FILE the INTERACTIVE loop instead (renderer.cc:338-615 without -b: frontend.h), its keyboard fed from a script
//               ("poll N", "down KEY", "up KEY", "tap KEY"; keys as on the reference's help screen: arrows, a z, s d f e, r, w q,
//               0-9, pgup pgdn, h, esc); --frame-ms T gives every frame T ms on the loop's clock (default: the measured time),
//               --no-brakes = configure --disable-brakes (raytraced modes are not frozen)
#include "renderer_host.h"
#include "frontend.h"

#include <fstream>
#include <sstream>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace mi355;

static void usage()
{
    fprintf(stderr,
            "Usage: render_cli [-b] [-r] [--bench] [-n frames] [-m mode] [-w] [-W width] [-H height] [-d device] [-g gpus] [-p in_flight] [--keep-canvas] [-o ppm_prefix] FILE\n"
            "  -m <mode>  1 points, 2 points from triangles, 3 wireframe, 4 ambient, 5 Gouraud, 6 Phong,\n"
            "             7 Phong+shadow maps, 8 Phong+soft shadow maps, 9 raytracing, 0 raytracing+AA\n"
            "  -w         use two lights        -n N  frames (default 100)\n"
            "  --keep-canvas  (modes 4-8) nothing but Scene::render* writes into the canvases: frames cross PCIe only where they differ from the last\n");
    exit(1);
}

static void write_ppm(const char *prefix, int frame, const Screen &canvas)
{
    char name[512];
    snprintf(name, sizeof name, "%s_%04d.ppm", prefix, frame);
    if (FILE *fp = fopen(name, "wb")) {
        fprintf(fp, "P6\n%d %d\n255\n", canvas._width, canvas._height);
        for (uint32_t p : canvas._pixels) { unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p}; fwrite(rgb, 1, 3, fp); }
        fclose(fp);
    }
}

// one `renderer -b -n frames` run; returns frames per second (time inside Scene::render* only, renderer.cc:584-585, 631-633)
static int g_depth = 0;            // --depth (0: the reference's 3)
static bool g_keepCanvas = false;  // --keep-canvas (Screen::_keepCanvas)

static double run(const char *fname, int mode, int frames, int W, int H, const std::vector<int> &devices, bool twoLights, bool periodic, const char *dump,
                  int inFlight = 1)
{
    Scene scene;
    if (g_depth > 0) scene._opts.max_ray_depth = g_depth;
    if (devices.size() > 1) scene._devices = devices;
    else if (!devices.empty()) scene._device = devices[0];
    Screen canvas(scene, W, H);
    canvas._keepCanvas = g_keepCanvas;
    scene.load(fname);
    printf("Vertexes: %zu Triangles: %zu\n", scene.numVertices(), scene.numTriangles());
    if (mode >= 9) scene.UpdateBoundingVolumeHierarchy(fname);          // renderer.cc:254-258 (untimed)
    const Vector3 lp = BenchmarkOrbit::lightPosition(), lp2 = BenchmarkOrbit::secondLightPosition();
    Light light(lp._x, lp._y, lp._z), light2(lp2._x, lp2._y, lp2._z);
    scene._lights.push_back(&light);
    if (twoLights) scene._lights.push_back(&light2);
    BenchmarkOrbit orbit;
    Camera sony(orbit.eye, Vector3(orbit.eye._x + 1.0f, orbit.eye._y, orbit.eye._z));
    for (Light *l : scene._lights) {                                    // renderer.cc:319-327 (untimed)
        l->CalculatePositionInCameraSpace(sony);
        l->RenderSceneIntoShadowBuffer(scene);
        l->CalculateXformFromWorldToLightSpace();
    }
    if (inFlight > 1) {
        // pipelined presentation: a ring of canvases, frame f waits for frame f - inFlight + 1 before it is shown
        std::vector<std::unique_ptr<Screen>> ring;
        for (int i = 0; i < inFlight; i++) {
            ring.emplace_back(new Screen(scene, W, H));                    // (page-locked by its first frame: direct DMA into the canvas)
            ring.back()->_keepCanvas = g_keepCanvas;
        }
        std::vector<int> ticket((size_t)inFlight, -1), frameOf((size_t)inFlight, 0);
        const int m = mode == 2 ? MI355_MODE_POINTS_FROM_TRIANGLES : mode;
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
            const int slot = f % inFlight;
            if (ticket[slot] >= 0) {                                       // the canvas's previous frame (f - inFlight): wait, present
                scene.renderWait(ticket[slot]); ring[slot]->ShowScreen(mode >= 9, true); ticket[slot] = -1;
                if (dump) write_ppm(dump, frameOf[slot], *ring[slot]);
            }
            {
                orbit.advance();
                sony.set(orbit.eye, orbit.lookat);
                if (mode >= 5) for (Light *l : scene._lights) l->CalculatePositionInCameraSpace(sony);
                if (mode >= 7) for (Light *l : scene._lights) l->CalculateXformFromCameraToLightSpace(sony);
                ticket[slot] = scene.renderAsync(m, sony, *ring[slot]);
                frameOf[slot] = f + 1;
            }
        }
        for (int k = 0; k < inFlight; k++) {                               // the last frames, oldest first
            const int i = (frames + k) % inFlight;
            if (ticket[i] >= 0) { scene.renderWait(ticket[i]); ring[i]->ShowScreen(mode >= 9, true); if (dump) write_ppm(dump, frameOf[i], *ring[i]); }
        }
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const double fps = frames / sec;
        printf("Rendering %d frames in %g seconds. (%g fps, %d in flight)\n", frames, sec, fps, inFlight);
        return fps;
    }
    double msSpentDrawing = 0, msAtLastReport = 0;
    int framesAtLastReport = 0;
    const auto tStart = std::chrono::steady_clock::now();
    double lastReport = 0;
    for (int f = 0; f < frames; f++) {
        orbit.advance();
        sony.set(orbit.eye, orbit.lookat);
        if (mode >= 5) for (Light *l : scene._lights) l->CalculatePositionInCameraSpace(sony);
        if (mode >= 7) for (Light *l : scene._lights) l->CalculateXformFromCameraToLightSpace(sony);
        const auto t0 = std::chrono::steady_clock::now();
        switch (mode) {
        case 1: scene.renderPoints(sony, canvas, false); break;
        case 2: scene.renderPoints(sony, canvas, true); break;
        case 3: scene.renderWireframe(sony, canvas); break;
        case 4: scene.renderAmbient(sony, canvas); break;
        case 5: scene.renderGouraud(sony, canvas); break;
        case 6: scene.renderPhong(sony, canvas); break;
        case 7: scene.renderPhongAndShadowed(sony, canvas); break;
        case 8: scene.renderPhongAndSoftShadowed(sony, canvas); break;
        case 9: scene.renderRaytracer(sony, canvas, false); break;
        default: scene.renderRaytracer(sony, canvas, true); break;
        }
        const auto t1 = std::chrono::steady_clock::now();
        msSpentDrawing += std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (periodic) {                                                  // renderer.cc:603-614: every 5 seconds of wall time
            const double wall = std::chrono::duration<double>(t1 - tStart).count();
            if (wall - lastReport >= 5.0) {
                printf("FPS: %g\n", (f + 1 - framesAtLastReport) / ((msSpentDrawing - msAtLastReport) / 1000.0));
                fflush(stdout);
                lastReport = wall; framesAtLastReport = f + 1; msAtLastReport = msSpentDrawing;
            }
        }
        if (dump) write_ppm(dump, f + 1, canvas);
    }
    const double fps = msSpentDrawing > 0 ? frames / (msSpentDrawing / 1000.0) : 0.0;
    if (msSpentDrawing > 0) printf("Rendering %d frames in %g seconds. (%g fps)\n", frames, msSpentDrawing / 1000.0, fps);
    return fps;
}

// the interactive loop on a key script; every drawn frame is shown (ShowScreen) and, with -o, dumped
static int runKeys(const char *fname, const char *keysFile, int mode, int W, int H, const std::vector<int> &devices, bool twoLights, long frameMS, bool brakes, const char *dump)
{
    std::ifstream in(keysFile);
    if (!in) throw std::string("cannot read key script ") + keysFile;
    std::stringstream text; text << in.rdbuf();
    Scene scene;
    if (devices.size() > 1) scene._devices = devices;
    else if (!devices.empty()) scene._device = devices[0];
    Screen canvas(scene, W, H);
    scene.load(fname);
    printf("Vertexes: %zu Triangles: %zu\n", scene.numVertices(), scene.numTriangles());
    scene.UpdateBoundingVolumeHierarchy(fname);        // (the reference builds it on the first raytraced frame, Raytracer.cc:797)
    int shown = 0;
    {
        FrontEnd fe(scene, &canvas, mode, twoLights, KeyScript(text.str()));
        fe.brakes = brakes;
        if (frameMS >= 0) fe.frameMS = [frameMS] { return frameMS; };
        fe.onFrame = [&](const FrontEnd::Frame &f) {
            shown++;
            if (dump) write_ppm(dump, shown, canvas);
            printf("frame %d: pass %llu mode %d eye %.9g %.9g %.9g%s\n", shown, (unsigned long long)f.pass, f.mode, f.eye._x, f.eye._y, f.eye._z, f.completed ? "" : " (abandoned)");
        };
        fe.run();
        printf("%d frames; %llu polls; last caption: %s\n", shown, (unsigned long long)fe.keys._polls, fe.caption.c_str());
        scene._lights.clear();
    }
    return 0;
}

int main(int argc, char **argv)
{
    int mode = 8, frames = -1, W = 800, H = 600;                   // defaults: renderer.cc:177-181, Defines.h:26-27
    const char *keysFile = nullptr; long frameMS = -1; bool brakes = true;
    std::vector<int> devices;
    bool twoLights = false, periodic = false, bench = false;
    int inFlight = 0;                                              // 0: not given
    const char *dump = nullptr, *fname = nullptr;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) usage(); return argv[++i]; };
        if (!strcmp(a, "-b")) continue;                            // always benchmarking: there is no UI
        else if (!strcmp(a, "-w")) twoLights = true;
        else if (!strcmp(a, "-r")) periodic = true;
        else if (!strcmp(a, "--bench")) bench = true;
        else if (!strcmp(a, "-n")) frames = atoi(next());
        else if (!strcmp(a, "-m")) { mode = atoi(next()); if (mode == 0) mode = 10; }
        else if (!strcmp(a, "-W")) W = atoi(next());
        else if (!strcmp(a, "-H")) H = atoi(next());
        else if (!strcmp(a, "-d")) devices.assign(1, atoi(next()));
        else if (!strcmp(a, "-g")) {
            const char *v = next();
            devices.clear();
            if (strchr(v, ',')) { for (const char *q = v; *q;) { devices.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; q++; } }
            else for (int d = 0; d < atoi(v); d++) devices.push_back(d);
            if (devices.empty()) usage();
        }
        else if (!strcmp(a, "-p")) { inFlight = atoi(next()); if (inFlight < 1 || inFlight > MI355_MAX_IN_FLIGHT) usage(); }
        else if (!strcmp(a, "-o")) dump = next();
        else if (!strcmp(a, "--keys")) keysFile = next();
        else if (!strcmp(a, "--frame-ms")) frameMS = atol(next());
        else if (!strcmp(a, "--no-brakes")) brakes = false;
        else if (!strcmp(a, "--keep-canvas")) g_keepCanvas = true;
        else if (!strcmp(a, "--depth")) { g_depth = atoi(next()); if (g_depth < 1 || g_depth > 4) usage(); }
        else if (a[0] == '-') usage();
        else fname = a;
    }
    if (!fname || mode < 1 || mode > 10) usage();
    if (frames < 0) frames = bench ? 500 : 100;
    // (frames in flight: one device, the tiled rasterizer and the raytracer; -r reports time inside the calls)
    // (kept canvases, rasterizer: four -- nothing is copied behind a frame, a fourth one in flight fills the gaps: 19.2 -> 21.7 k fps at 1080p)
    if (inFlight == 0) inFlight = (devices.size() > 1 || periodic || mode < 4) ? 1 : (g_keepCanvas && mode <= 8 && MI355_MAX_IN_FLIGHT >= 4 ? 4 : 3);
    try {
        if (keysFile) return runKeys(fname, keysFile, mode, W, H, devices, twoLights, frameMS, brakes, dump);
        if (!bench) { run(fname, mode, frames, W, H, devices, twoLights, periodic, dump, inFlight); return 0; }
        // src/Makefile.am:25-26: five runs, then the statistics of their frame rates
        std::vector<double> fps;
        for (int i = 0; i < 5; i++) fps.push_back(run(fname, mode, frames, W, H, devices, twoLights, periodic, nullptr, inFlight));
        // (the perl of src/Makefile.am:26: sample variance, the middle element of the sorted rates, "%15s: %f")
        double total = 0, totalSq = 0;
        for (double v : fps) { printf("%g\n", v); total += v; totalSq += v * v; }
        const double n = (double)fps.size(), variance = (totalSq - total * total / n) / (n - 1);
        std::vector<double> sorted = fps; std::sort(sorted.begin(), sorted.end());
        size_t len = sorted.size(); if (len % 2) len++;
        const char *names[5] = {"Average value", "Std deviation", "Median", "Min", "Max"};
        const double vals[5] = {total / n, std::sqrt(variance > 0 ? variance : 0), sorted[len / 2 - 1], sorted.front(), sorted.back()};
        for (int i = 0; i < 5; i++) printf("%15s: %f\n", names[i], vals[i]);
    } catch (const std::string &s) {
        fprintf(stderr, "%s\n", s.c_str());
        return 1;
    }
    return 0;
}
