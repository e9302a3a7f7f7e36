// Source-compatibility forwarder for callers that include the reference's header name
// (reference src/PathtracingUtility/scene_loader.h); the declarations live in ptk_rt/utility.hpp.
#pragma once
#include "ptk_rt/utility.hpp"
