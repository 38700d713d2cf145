// vec.h -- main.cpp reaches headers/util.h's three-argument max through this include (main.cpp:106).
#pragma once

#include <algorithm>

using std::max;
using std::min;
template <class T>
inline T max(T a, T b, T c) { return std::max(a, std::max(b, c)); }
template <class T>
inline T min(T a, T b, T c) { return std::min(a, std::min(b, c)); }
