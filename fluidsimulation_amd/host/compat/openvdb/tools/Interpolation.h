// openvdb/tools/Interpolation.h -- everything the FLIP path needs from OpenVDB's tools lives in flipvdb.h
#pragma once
#include "../openvdb.h"
