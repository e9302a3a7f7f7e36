// Source-compatibility forwarder for callers that include the reference's header name
// (reference src/Pathtracing/sampler.h); the declarations live in ptk_rt/samplers.hpp.
#pragma once
#include "ptk_rt/samplers.hpp"
