// SFML/Graphics.hpp -- headless stand-in for the window of main.cpp:150-162: there is no display on a GPU node.
// The window reports itself open for FLIP_HEADLESS_FRAMES polls of isOpen() (default: for ever, i.e. until main.cpp's
// own frame counter ends the loop); drawing calls do nothing.
#pragma once

#include <cstdlib>
#include <string>

namespace sf {
struct Color {
    unsigned char r = 0, g = 0, b = 0, a = 255;
    Color() {}
    Color(unsigned char r_, unsigned char g_, unsigned char b_, unsigned char a_ = 255): r(r_), g(g_), b(b_), a(a_) {}
    static const Color Black, White, Red, Green, Blue;
};
inline const Color Color::Black(0, 0, 0), Color::White(255, 255, 255), Color::Red(255, 0, 0), Color::Green(0, 255, 0), Color::Blue(0, 0, 255);
struct Vector2f {
    float x = 0, y = 0;
    Vector2f() {}
    Vector2f(float x_, float y_): x(x_), y(y_) {}
};
struct VideoMode {
    unsigned width, height;
    VideoMode(unsigned w, unsigned h): width(w), height(h) {}
};
struct Event {
    enum EventType { Closed, Other };
    EventType type = Other;
};
class RenderWindow {
public:
    RenderWindow(VideoMode, const std::string &) {
        if (const char *e = std::getenv("FLIP_HEADLESS_FRAMES")) _frames_left = std::atol(e);
    }
    bool isOpen() {
        if (_frames_left < 0) return _open;
        return _open && _frames_left-- > 0;
    }
    bool pollEvent(Event &) { return false; }
    void close() { _open = false; }
    void clear(const Color & = Color::Black) {}
    void display() {}
private:
    bool _open = true;
    long _frames_left = -1;
};
} // namespace sf
