// openvdb/openvdb.h -- stand-in for the OpenVDB umbrella header where OpenVDB is not installed: the names the
// reference's main.cpp and public headers spell with openvdb:: resolve to the dense-grid vocabulary of flipvdb.h.
// With OpenVDB installed this directory simply stays off the include path.
#pragma once

#include "flipvdb.h"

#include <chrono>
#include <cstdio>
#include <iostream>

namespace openvdb = flipvdb;
