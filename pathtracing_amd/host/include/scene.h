// Source-compatibility forwarder for callers that include the reference's header name
// (reference src/Pathtracing/scene.h); the declarations live in ptk_rt/scene_graph.hpp.
#pragma once
#include "ptk_rt/scene_graph.hpp"
