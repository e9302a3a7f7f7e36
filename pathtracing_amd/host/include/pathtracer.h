// Source-compatibility forwarder for callers that include the reference's header name
// (reference src/Pathtracing/pathtracer.h); the declarations live in ptk_rt/render.hpp.
#pragma once
#include "ptk_rt/render.hpp"
