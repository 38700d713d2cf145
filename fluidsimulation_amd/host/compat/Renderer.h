// Renderer.h -- headless stand-in for headers/Renderer.h:17-38 (an SFML slice view, out of the path): main.cpp:137-138,161
// construct it, initialise it and call draw() once per frame.
#pragma once

#include "FluidDomain.h"
#include "render_utils.h"

#include <SFML/Graphics.hpp>

#define WINDOW_SIZE 800

class Renderer {
public:
    Renderer(int sim_size_x, int sim_size_y): _nx(sim_size_x), _ny(sim_size_y) {}
    void initialisation() {}
    int widthX() { return _nx; }
    int widthY() { return _ny; }
    void draw(FluidDomain &, sf::RenderWindow &) {}
private:
    int _nx, _ny;
};
