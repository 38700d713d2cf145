// render_utils.h -- main.cpp includes it for the renderer's helpers; the headless build needs none of them.
#pragma once
