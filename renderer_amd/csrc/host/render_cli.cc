// render_cli.cc -- headless counterpart of `renderer -b -n N -m MODE FILE` (renderer.cc:168-642).
//
// Reproduces the reference's benchmark loop -- same light and eye placement, same auto-spin
// camera sequence, BVH build and shadow-map generation outside the timed region, fps =
// frames / seconds spent inside Scene::render* -- on top of the C++ host API, i.e. on the GPU.
// There is no window: -o PREFIX dumps frames as binary PPM instead of SDL_Flip.
#include "renderer_host.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

using namespace mi355;

static void usage()
{
    fprintf(stderr,
            "Usage: render_cli [-b] [-n frames] [-m mode] [-w] [-W width] [-H height] [-d device] [-o ppm_prefix] FILE\n"
            "  -m <mode>  1 points, 2 points from triangles, 4 ambient, 5 Gouraud, 6 Phong,\n"
            "             7 Phong+shadow maps, 8 Phong+soft shadow maps, 9 raytracing, 0 raytracing+AA\n"
            "  -w         use two lights        -n N  frames (default 100)\n");
    exit(1);
}

int main(int argc, char **argv)
{
    int mode = 8, frames = 100, W = 800, H = 600, device = 0;      // defaults: renderer.cc:177-181, Defines.h:26-27
    bool twoLights = false;
    const char *dump = nullptr, *fname = nullptr;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) usage(); return argv[++i]; };
        if (!strcmp(a, "-b")) continue;                            // always benchmarking: there is no UI
        else if (!strcmp(a, "-w")) twoLights = true;
        else if (!strcmp(a, "-n")) frames = atoi(next());
        else if (!strcmp(a, "-m")) { mode = atoi(next()); if (mode == 0) mode = 10; }
        else if (!strcmp(a, "-W")) W = atoi(next());
        else if (!strcmp(a, "-H")) H = atoi(next());
        else if (!strcmp(a, "-d")) device = atoi(next());
        else if (!strcmp(a, "-o")) dump = next();
        else if (a[0] == '-') usage();
        else fname = a;
    }
    if (!fname || mode < 1 || mode > 10 || mode == 3) usage();
    try {
        Scene scene;
        scene._device = device;
        Screen canvas(scene, W, H);
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
        double msSpentDrawing = 0;
        for (int f = 0; f < frames; f++) {
            orbit.advance();
            sony.set(orbit.eye, orbit.lookat);
            if (mode >= 5) for (Light *l : scene._lights) l->CalculatePositionInCameraSpace(sony);
            if (mode >= 7) for (Light *l : scene._lights) l->CalculateXformFromCameraToLightSpace(sony);
            const auto t0 = std::chrono::steady_clock::now();
            switch (mode) {
            case 1: scene.renderPoints(sony, canvas, false); break;
            case 2: scene.renderPoints(sony, canvas, true); break;
            case 4: scene.renderAmbient(sony, canvas); break;
            case 5: scene.renderGouraud(sony, canvas); break;
            case 6: scene.renderPhong(sony, canvas); break;
            case 7: scene.renderPhongAndShadowed(sony, canvas); break;
            case 8: scene.renderPhongAndSoftShadowed(sony, canvas); break;
            case 9: scene.renderRaytracer(sony, canvas, false); break;
            default: scene.renderRaytracer(sony, canvas, true); break;
            }
            msSpentDrawing += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (dump) {
                char name[512];
                snprintf(name, sizeof name, "%s_%04d.ppm", dump, f + 1);
                if (FILE *fp = fopen(name, "wb")) {
                    fprintf(fp, "P6\n%d %d\n255\n", W, H);
                    for (uint32_t p : canvas._pixels) { unsigned char rgb[3] = {(unsigned char)(p >> 16), (unsigned char)(p >> 8), (unsigned char)p}; fwrite(rgb, 1, 3, fp); }
                    fclose(fp);
                }
            }
        }
        if (msSpentDrawing > 0)
            printf("Rendering %d frames in %g seconds. (%g fps)\n", frames, msSpentDrawing / 1000.0, frames / (msSpentDrawing / 1000.0));
    } catch (const std::string &s) {
        fprintf(stderr, "%s\n", s.c_str());
        return 1;
    }
    return 0;
}
