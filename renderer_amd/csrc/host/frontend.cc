// frontend.cc -- see frontend.h.  Every block cites the lines of renderer.cc / Keyboard.cc / Raytracer.cc it restates; the float
// arithmetic keeps their operand order and types (coord = float, the trigonometry in float, DEGREES_TO_RADIANS through double).
#include "frontend.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <sstream>

namespace mi355 {

namespace {
inline coord degToRad(double x) { return (coord)(x * M_PI / 180.0); }     // renderer.cc:95 (x arrives as a float there)
const char *const modeNames[10] = {"Points", "Points from triangles", "Lines", "Ambient", "Gouraud", "Phong", "Phong with shadow maps",
                                   "Phong with soft shadow maps", "Raytracing", "Raytracing with antialiasing"};
long nowMS() { return (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

Key keyFromName(const std::string &n)
{
    static const struct { const char *name; Key k; } t[] = {
        {"up", KEY_UP}, {"down", KEY_DOWN}, {"left", KEY_LEFT}, {"right", KEY_RIGHT}, {"a", KEY_A}, {"z", KEY_Z}, {"w", KEY_W}, {"q", KEY_Q},
        {"s", KEY_S}, {"d", KEY_D}, {"f", KEY_F}, {"e", KEY_E}, {"r", KEY_R}, {"h", KEY_H}, {"esc", KEY_ESCAPE}, {"escape", KEY_ESCAPE},
        {"pgdn", KEY_PAGEDOWN}, {"pagedown", KEY_PAGEDOWN}, {"pgup", KEY_PAGEUP}, {"pageup", KEY_PAGEUP}, {"0", KEY_0}, {"1", KEY_1}, {"2", KEY_2},
        {"3", KEY_3}, {"4", KEY_4}, {"5", KEY_5}, {"6", KEY_6}, {"7", KEY_7}, {"8", KEY_8}, {"9", KEY_9}};
    for (const auto &e : t) if (n == e.name) return e.k;
    return KEY_NONE;
}

// Keyboard.cc:35-119: at most one event per call; a key-down sets its flag, a key-up clears it
void Keyboard::poll(bool)
{
    _polls++;
    if (!source) return;
    const KeyEvent ev = source();
    if (ev.type == KeyEvent::QUIT) { _quit = true; return; }
    if (ev.type != KeyEvent::DOWN && ev.type != KeyEvent::UP) return;
    const uint32_t v = ev.type == KeyEvent::DOWN ? 1u : 0u;
    switch (ev.key) {
    case KEY_UP: _isUp = v; break;            case KEY_DOWN: _isDown = v; break;
    case KEY_LEFT: _isLeft = v; break;        case KEY_RIGHT: _isRight = v; break;
    case KEY_A: _isForward = v; break;        case KEY_Z: _isBackward = v; break;
    case KEY_W: _isLight = v; break;          case KEY_Q: _isLight2 = v; break;
    case KEY_S: _isS = v; break;              case KEY_D: _isD = v; break;
    case KEY_F: _isF = v; break;              case KEY_E: _isE = v; break;
    case KEY_R: _isR = v; break;              case KEY_H: _isH = v; break;
    case KEY_ESCAPE: _isAbort = v; break;
    case KEY_PAGEDOWN: _isPgDown = v; break;  case KEY_PAGEUP: _isPgUp = v; break;
    case KEY_0: _is0 = v; break; case KEY_1: _is1 = v; break; case KEY_2: _is2 = v; break; case KEY_3: _is3 = v; break; case KEY_4: _is4 = v; break;
    case KEY_5: _is5 = v; break; case KEY_6: _is6 = v; break; case KEY_7: _is7 = v; break; case KEY_8: _is8 = v; break; case KEY_9: _is9 = v; break;
    default: break;
    }
}

KeyScript::KeyScript(const std::string &text)
{
    std::istringstream in(text);
    std::string line;
    int lineNo = 0;
    while (std::getline(in, line)) {
        lineNo++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ls(line);
        std::string verb, arg;
        if (!(ls >> verb)) continue;
        ls >> arg;
        if (verb == "poll") {
            const long n = atol(arg.c_str());
            if (n < 0 || n > 100000000L) throw std::string("key script line ") + std::to_string(lineNo) + ": bad count";
            events.insert(events.end(), (size_t)n, KeyEvent());
            continue;
        }
        const Key k = keyFromName(arg);
        if (k == KEY_NONE || (verb != "down" && verb != "up" && verb != "tap"))
            throw std::string("key script line ") + std::to_string(lineNo) + ": expected poll N | down KEY | up KEY | tap KEY";
        KeyEvent e; e.key = k;
        if (verb != "up") { e.type = KeyEvent::DOWN; events.push_back(e); }
        if (verb != "down") { e.type = KeyEvent::UP; events.push_back(e); }
    }
}

KeyEvent KeyScript::operator()()
{
    if (next < events.size()) return events[next++];
    KeyEvent q; q.type = KeyEvent::QUIT;
    return q;
}

// renderer.cc:246-336: the state before the first pass of the loop
FrontEnd::FrontEnd(Scene &sc, Screen *cv, int m, bool two, std::function<KeyEvent()> source)
    : scene(sc), canvas(cv), mode(m), useTwoLights(two), angle1(0.0f), angle2((coord)(0.0f * M_PI / 180.f)), angle3((coord)(45.0f * M_PI / 180.f)),
      dAngle(degToRad(0.3f)), eye(Scene::MaxCoordAfterRescale * 4.0f, 0.0f, 0.0f), lookat(), sony(1.f, 0.f, 0.f, 0.f, 0.f, 0.f),
      light(4.0f * Scene::MaxCoordAfterRescale, 4.0f * Scene::MaxCoordAfterRescale, 4.0f * Scene::MaxCoordAfterRescale),
      light2(4.0f * Scene::MaxCoordAfterRescale, -4.0f * Scene::MaxCoordAfterRescale, 4.0f * Scene::MaxCoordAfterRescale),
      oldEyePosition(1e10f, 1e10f, 1e10f), oldLookAtPosition(1e10f, 1e10f, 1e10f), oldLightPosition(1e10f, 1e10f, 1e10f)
{
    keys.source = std::move(source);
    const coord maxi = Scene::MaxCoordAfterRescale, LightDistanceFactor = 4.0f;
    scene._lights.push_back(&light);                                       // :278-285
    light._x = LightDistanceFactor * maxi * cosf(angle3);
    light._y = LightDistanceFactor * maxi * sinf(angle3);
    light.ClearShadowBuffer();
    if (useTwoLights) { scene._lights.push_back(&light2); light2.ClearShadowBuffer(); }   // :288-296
    lookat = Vector3(eye._x + 1.0f * cosf(angle2) * cosf(angle1), eye._y + 1.0f * cosf(angle2) * sinf(angle1), eye._z + 1.0f * sinf(angle2));   // :301-303
    sony.set(eye, lookat);
    caption = modeNames[mode - 1];                                         // :308
    keys.poll();                                                           // :318
    for (Light *l : scene._lights) {                                       // :319-327
        l->CalculatePositionInCameraSpace(sony);
        if (canvas) l->RenderSceneIntoShadowBuffer(scene);
        l->CalculateXformFromWorldToLightSpace();
    }
}

void FrontEnd::relight()
{
    light.ClearShadowBuffer();
    if (canvas) light.RenderSceneIntoShadowBuffer(scene);
    dirtyShadowBuffer = false;
}

// renderer.cc:141-163, the picture left out: wait for H (or ESC) to be pressed, then for both to be up again
void FrontEnd::showHelp()
{
    keys.poll();
    while (!keys._isH && !keys._isAbort && !keys._quit) keys.poll();
    while ((keys._isH || keys._isAbort) && !keys._quit) keys.poll();
}

// Raytracer.cc:791-868 with HANDLERAYTRACER: the reference traces scanline by scanline, polls the keyboard once after each, gives
// up when ESC is down (after waiting for its release) and shows the buffer every 16 scanlines.  Here the scanlines are traced 16
// at a time -- Scene::renderRaytracerRows: one band-sharded frame call per band, rendered compactly and copied into the canvas's
// own rows, the other rows keep what they hold -- and the polls of a band's scanlines follow it, so an abort takes effect at the
// next multiple of 16 scanlines at the latest.
// The keyboard is the CALL's OWN, as in the reference (`Keyboard keys;`, Raytracer.cc:812): it starts with every flag clear and
// shares only the event queue (and the window's close request, on which the reference exits the process) with the loop's
// keyboard -- a key that goes down during the frame is consumed here and never reaches the loop's flags, a key that goes up
// during the frame leaves the loop's flag set.
bool FrontEnd::renderRaytracerWithBrakes(bool antialias)
{
    const int H = canvas ? canvas->_height : 600;
    Keyboard k;
    k.source = [this] { return keys.source ? keys.source() : KeyEvent(); };
    k._quit = keys._quit;
    struct Merge { Keyboard &loop, &mine; ~Merge() { loop._quit = loop._quit || mine._quit; loop._polls += mine._polls; } } merge{keys, k};
    bool completed = true;
    for (int y0 = 0; y0 < H && completed; y0 += 16) {
        if (canvas) scene.renderRaytracerRows(sony, *canvas, antialias, y0, 16);
        for (int y = y0; y < std::min(H, y0 + 16); y++) {
            k.poll(false);                                                 // :842
            if (k._isAbort || k._quit) {
                while (k._isAbort && !k._quit) k.poll(false);              // :844
                completed = false;
                break;
            }
            if (15 == (y & 15)) {                                          // :852-862
                std::ostringstream percentage;
                percentage << (antialias ? "Anti-aliased r" : "R") << "aytracing... hit ESCAPE to abort (" << int(100. * y / H) << "%)";
                caption = percentage.str();
                if (canvas) canvas->ShowScreen(true, false);
            }
        }
    }
    if (completed && canvas) canvas->ShowScreen(true, true);               // :866
    return completed;
}

bool FrontEnd::step()
{
    Keyboard &k = keys;
    if (k._isAbort || k._quit) return false;                               // renderer.cc:338
    pass++;
    const coord maxi = Scene::MaxCoordAfterRescale, LightDistanceFactor = 4.0f;
    if (k._isH) {                                                          // :342-351
        while (k._isH && !k._quit) k.poll();
        showHelp();
        msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true;
        return true;
    }
    if (k._isLeft) angle1 -= dAngle;                                       // :352-359
    if (k._isRight) angle1 += dAngle;
    if (k._isUp) angle2 = std::min(angle2 + dAngle, degToRad(89.0f));
    if (k._isDown) angle2 = std::max(angle2 - dAngle, degToRad(-89.0f));
    if (k._isForward || k._isBackward) {                                   // :360-376
        Vector3 fromEyeToLookat(lookat);
        fromEyeToLookat -= eye;
        if (autoRotate) fromEyeToLookat *= 0.05f;
        else fromEyeToLookat *= 0.05f * maxi;
        if (k._isForward) eye += fromEyeToLookat;
        else eye -= fromEyeToLookat;
    }
    if (k._isS || k._isF || k._isE || k._isD) {                            // :377-390
        Vector3 eyeToLookatPoint = lookat;
        eyeToLookatPoint -= eye;
        eyeToLookatPoint.normalize();
        const Vector3 zenith(0.f, 0.f, 1.f);
        Vector3 rightAxis = cross(eyeToLookatPoint, zenith);
        rightAxis.normalize();
        Vector3 upAxis = cross(rightAxis, eyeToLookatPoint);
        upAxis.normalize();
        if (k._isS) { rightAxis *= 0.05f * maxi; eye -= rightAxis; }
        if (k._isF) { rightAxis *= 0.05f * maxi; eye += rightAxis; }      // (a second scaling when S is down too, as there)
        if (k._isD) { upAxis *= 0.05f * maxi; eye -= upAxis; }
        if (k._isE) { upAxis *= 0.05f * maxi; eye += upAxis; }
    }
    if (k._isR) {                                                          // :391-410
        while (k._isR && !k._quit) k.poll();
        autoRotate = !autoRotate;
        if (!autoRotate) {
            Vector3 eyeToAxes = eye;
            eyeToAxes.normalize();
            angle2 = asinf(-eyeToAxes._z);
            angle1 = (eye._y < 0) ? acosf(eyeToAxes._x / cosf(angle2)) : -acosf(eyeToAxes._x / cosf(angle2));
        } else {
            angle1 = -angle1;
            angle2 = -angle2;
        }
    }
    if (k._isLight || k._isLight2) {                                       // :411-431
        if (k._isLight) angle3 += 4 * dAngle;
        else angle3 -= 4 * dAngle;
        light._x = LightDistanceFactor * maxi * cosf(angle3);
        light._y = LightDistanceFactor * maxi * sinf(angle3);
        dirtyShadowBuffer = true;
        if (mode == 7 || mode == 8) relight();
        else if (mode == 9 || mode == 10) light.CalculateXformFromWorldToLightSpace();
    }
    bool newMode = false;                                                  // :432-451
    if (k._is0 || k._is1 || k._is2 || k._is3 || k._is4 || k._is5 || k._is6 || k._is7 || k._is8 || k._is9) {
        if (k._is1) mode = 1;
        if (k._is2) mode = 2;
        if (k._is3) mode = 3;
        if (k._is4) mode = 4;
        if (k._is5) mode = 5;
        if (k._is6) mode = 6;
        if (k._is7) mode = 7;
        if (k._is8) mode = 8;
        if (k._is9) mode = 9;
        if (k._is0) mode = 10;
        while ((k._is0 || k._is1 || k._is2 || k._is3 || k._is4 || k._is5 || k._is6 || k._is7 || k._is8 || k._is9) && !k._quit) k.poll();
        newMode = true;
    }
    if (k._isPgDown || k._isPgUp) {                                        // :452-461
        const uint32_t up = k._isPgUp;
        while ((k._isPgDown || k._isPgUp) && !k._quit) k.poll();
        if (!up) mode = mode == 1 ? 10 : mode - 1;
        else mode = mode == 10 ? 1 : mode + 1;
        newMode = true;
    }
    if (newMode) {                                                         // :462-479
        caption = modeNames[mode - 1];
        if (dirtyShadowBuffer && (mode == 7 || mode == 8)) relight();
        if (mode == 9 || mode == 10) light.CalculateXformFromWorldToLightSpace();
        dAngle = degToRad(0.3f);
        msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true;
        return true;
    }
    if (!autoRotate) {                                                     // :481-495
        lookat._x = eye._x - 1.0f * cosf(angle2) * cosf(angle1);
        lookat._y = eye._y + 1.0f * cosf(angle2) * sinf(angle1);
        lookat._z = eye._z + 1.0f * sinf(angle2);
    } else {
        angle1 -= dAngle;
        lookat._x = 0; lookat._y = 0; lookat._z = 0;
        const coord distance = sqrtf(eye._x * eye._x + eye._y * eye._y + eye._z * eye._z);
        eye._x = distance * cosf(angle2) * cosf(angle1);
        eye._y = distance * cosf(angle2) * sinf(angle1);
        eye._z = distance * sinf(angle2);
    }
    sony.set(eye, lookat);                                                 // :497
    if (mode >= 5) {                                                       // :499-508
        light.CalculatePositionInCameraSpace(sony);
        if (useTwoLights) light2.CalculatePositionInCameraSpace(sony);
    }
    if (mode >= 7) {
        light.CalculateXformFromCameraToLightSpace(sony);
        if (useTwoLights) light2.CalculateXformFromCameraToLightSpace(sony);
    }
    if (oldLightPosition != Vector3(light) || oldEyePosition != eye || oldLookAtPosition != lookat || forceRedraw) {       // :511-586
        oldLightPosition = Vector3(light); oldEyePosition = eye; oldLookAtPosition = lookat;
        forceRedraw = false;
        _t0 = nowMS();
        bool completed = true, backToSoftShadows = false;
        if (canvas || mode >= 9)
            switch (mode) {
            case 1: scene.renderPoints(sony, *canvas, false); break;
            case 2: scene.renderPoints(sony, *canvas, true); break;
            case 3: scene.renderWireframe(sony, *canvas); break;
            case 4: scene.renderAmbient(sony, *canvas); break;
            case 5: scene.renderGouraud(sony, *canvas); break;
            case 6: scene.renderPhong(sony, *canvas); break;
            case 7: scene.renderPhongAndShadowed(sony, *canvas); break;
            case 8: scene.renderPhongAndSoftShadowed(sony, *canvas); break;
            default:
                if (brakes) {                                              // :553-573: the "freeze frame" handling of the raytraced modes
                    const long t0 = nowMS();
                    completed = renderRaytracerWithBrakes(mode == 10);
                    if (completed) {
                        std::ostringstream msg;
                        msg << (mode == 10 ? "Anti-aliased r" : "R") << "aytracing completed in " << (nowMS() - t0 + 999) / 1000
                            << " seconds - hit ESC to return to soft shadowmapping mode...";
                        caption = msg.str();
                        while (!k._isAbort && !k._quit) k.poll();
                        while (k._isAbort && !k._quit) k.poll();
                    }
                    backToSoftShadows = true;
                } else if (canvas) scene.renderRaytracer(sony, *canvas, mode == 10);
                break;
            }
        if (onFrame) {
            Frame f{pass, mode, eye, lookat, Vector3(light), sony._mv, dAngle, autoRotate, completed};
            onFrame(f);
        }
        if (backToSoftShadows) {                                           // :566-572
            mode = 8;
            caption = modeNames[mode - 1];
            msSpentDrawing = 0; framesDrawn = 0; forceRedraw = true;
            return true;
        }
        framesDrawn++;                                                     // :584-585
        msSpentDrawing += frameMS ? frameMS() : nowMS() - _t0;
    }
    k.poll();                                                              // :587
    if (msSpentDrawing)                                                    // :597-601
        dAngle += (degToRad(9.0f / (framesDrawn / (msSpentDrawing / 1000.0f))) - dAngle) / 15.0f;
    return true;
}

} // namespace mi355
