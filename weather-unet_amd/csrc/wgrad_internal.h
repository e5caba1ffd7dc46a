// internal: production weight-gradient path (conv3x3_wgrad_v2.hip), dispatched from wu_conv3x3_wgrad
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct WgradV2Plan { int mode, tiles_x, tiles_y, ntiles, tiles_per_split, splits, co_blocks, ci_blocks; size_t ws; };
bool wgrad_v2_eligible(int H, int W, int Cin, int Cout, int stride, int dtype, bool gated);
WgradV2Plan wgrad_v2_plan(int N, int H, int W, int Cin, int Cout);
int wgrad_v2_launch(const void* x, int ldx, const void* dy, int lddy, float* slab, float* bslab,
                    int N, int H, int W, int Cin, int Cout, const WgradV2Plan& p, hipStream_t s);
