// frontend.h -- the reference's interactive frame loop (renderer.cc:243-642) and its Keyboard (Keyboard.h:27-38,
// Keyboard.cc:30-119) without a window system.
//
// The reference's front-end is SDL: a window, SDL_PollEvent, a title bar.  SDL is not what the loop IS, though: it is a state
// machine from key flags to eye / lookat / light angle / render mode, one pass of `while(!keys._isAbort)` per tick, and that part
// needs no SDL.  FrontEnd::step() is one pass of that loop, operation for operation; Keyboard::poll() takes its events from a
// caller-supplied source where the reference calls SDL_PollEvent (at most ONE event per poll, as there); a binding for a real
// window system only has to feed that source and blit Screen::_pixels in Screen::_present.  `render_cli --keys FILE` drives it
// from a script.  The clock is injectable too (the loop adapts its rotation step to the measured frame rate,
// renderer.cc:597-601): scripted runs use a virtual one so that they are reproducible.
#pragma once

#include <functional>
#include <string>
#include <vector>

#include "renderer_host.h"

namespace mi355 {

enum Key {     // the keys Keyboard.cc:43-69 knows
    KEY_NONE = 0, KEY_UP, KEY_DOWN, KEY_LEFT, KEY_RIGHT, KEY_A, KEY_Z, KEY_W, KEY_Q, KEY_S, KEY_D, KEY_F, KEY_E, KEY_R, KEY_H, KEY_ESCAPE,
    KEY_PAGEDOWN, KEY_PAGEUP, KEY_0, KEY_1, KEY_2, KEY_3, KEY_4, KEY_5, KEY_6, KEY_7, KEY_8, KEY_9, KEY_COUNT
};
Key keyFromName(const std::string &name);          // "up", "a", "esc", "pgup", "7", ...

struct KeyEvent { enum Type { NONE = 0, DOWN, UP, QUIT } type = NONE; Key key = KEY_NONE; };

struct Keyboard {                                  // Keyboard.h:27-38
    uint32_t _isDown = 0, _isUp = 0, _isLeft = 0, _isRight = 0;
    uint32_t _isForward = 0, _isBackward = 0, _isLight = 0, _isLight2 = 0;
    uint32_t _isAbort = 0, _isPgUp = 0, _isPgDown = 0;
    uint32_t _isS = 0, _isD = 0, _isE = 0, _isF = 0, _isR = 0, _isH = 0;
    uint32_t _is0 = 0, _is1 = 0, _is2 = 0, _is3 = 0, _is4 = 0, _is5 = 0, _is6 = 0, _is7 = 0, _is8 = 0, _is9 = 0;
    bool _quit = false;                            // SDL_QUIT arrived (the reference exits the process, Keyboard.cc:111-113)
    uint64_t _polls = 0;
    std::function<KeyEvent()> source;              // SDL_PollEvent's role: the next pending event, or NONE
    void poll(bool bYield = true);                 // Keyboard.cc:35-119
};

// A scripted event source: "poll N" = the next N polls find nothing, "down KEY" / "up KEY" = the next poll finds this event,
// "tap KEY" = down then up; when the script is used up every further poll finds QUIT.  '#' starts a comment.
struct KeyScript {
    std::vector<KeyEvent> events;                  // NONE entries = polls that find nothing
    size_t next = 0;
    explicit KeyScript(const std::string &text);   // throws std::string on a line it cannot read
    KeyEvent operator()();
};

struct FrontEnd {
    // one drawn frame, for whoever watches (tests compare these with the oracle's restatement of the same loop)
    struct Frame { uint64_t pass; int mode; Vector3 eye, lookat, light; Matrix3 mv; coord dAngle; bool autoRotate; bool completed; };

    Scene &scene;
    Screen *canvas;                                // NULL: a dry run -- the state machine alone, nothing is rendered
    Keyboard keys;
    int mode;
    bool useTwoLights;
    bool brakes = true;                            // HANDLERAYTRACER (configure --disable-brakes turns it off, renderer.cc:540-566)
    bool autoRotate = true;
    coord angle1, angle2, angle3, dAngle;
    Vector3 eye, lookat;
    Camera sony;
    Light light, light2;
    uint32_t framesDrawn = 0;
    long msSpentDrawing = 0;
    bool dirtyShadowBuffer = true, forceRedraw = false;
    Vector3 oldEyePosition, oldLookAtPosition, oldLightPosition;
    uint64_t pass = 0;
    std::string caption;                           // what SDL_WM_SetCaption would show
    std::function<long()> frameMS;                 // milliseconds the frame just drawn took (default: measured)
    std::function<void(const Frame &)> onFrame;

    FrontEnd(Scene &scene, Screen *canvas, int mode, bool useTwoLights, std::function<KeyEvent()> source);
    bool step();                                   // one pass of the loop of renderer.cc:338-615; false: the loop has ended
    void run() { while (step()) {} }

private:
    long _t0 = 0;
    bool renderRaytracerWithBrakes(bool antialias);   // Raytracer.cc:791-868 with HANDLERAYTRACER: abort between scanlines
    void showHelp();                                  // renderer.cc:141-163: the key protocol of the help screen
    void relight();
};

} // namespace mi355
