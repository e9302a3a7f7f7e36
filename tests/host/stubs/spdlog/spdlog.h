// TEST-ONLY stand-in for spdlog (an un-vendored submodule of the reference, .gitmodules:10-12): no-op
// info / warn / error, so that tests/test_host_cpp.py can SYNTAX-CHECK the reference's callers
// (src/CLI/main.cpp, src/PathtracingUtility/*.cpp) against the drop-in headers and LINK-CHECK the CLI against the drop-in
// library.  Nothing built with it is ever run, it is not part of the product, and it is not used to build an oracle.
#pragma once
namespace spdlog {
template <typename... A> inline void info(const A&...) {}
template <typename... A> inline void warn(const A&...) {}
template <typename... A> inline void error(const A&...) {}
} // namespace spdlog
