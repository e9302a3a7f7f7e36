// compat forwarder: everything lives in glm.hpp (see its header comment).
#pragma once
#include "../glm.hpp"
