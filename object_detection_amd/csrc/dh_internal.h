// Entry points shared between translation units of libdethead (not part of the C ABI).
#pragma once
#include <stddef.h>
#include <stdint.h>

int dh_launch_build_targets(const float* targets, int64_t nt, const float* anchors, int nl, int na, const int* ny, const int* nx,
                            double anchor_t, int64_t cap, int64_t* idx, float* tbox, float* anch, int64_t* tcls, int* counts,
                            void* workspace, size_t workspace_bytes, void* fill, size_t fill_bytes, void* stream);
