// Per-kernel-family launch timing (lsn_prof_*): HIP events recorded on the launch stream around each launch of a
// family, so bench.py can quote every family's own average duration, algorithmic flops and bytes live and put the
// roofline on whichever family dominates the step.  State lives in dcn.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace lsn {

enum {
    PROF_FWD = 0,         // deformable conv forward
    PROF_BWD_DATA = 1,    // deformable conv backward-data (bin / scan / fill / sort / GEMM / gather)
    PROF_WGRAD = 2,       // deformable conv weight gradient
    PROF_CONV_FWD = 3,    // dense conv forward (conv_mm_kernel)
    PROF_CONV_BWD_DATA = 4,   // dense conv data gradient (the same kernel on grad_output, per residue class)
    PROF_CONV_WGRAD = 5,  // dense conv weight / bias gradient
    PROF_NORM = 6,        // frozen-BN + add + ReLU and GroupNorm passes (HBM-bound streaming kernels)
    PROF_GCONV = 7,       // grouped conv (ResNeXt)
    PROF_N = 8
};

struct ProfRec {
    hipEvent_t e0, e1;
    int fam;
    double flops, bytes;
};

bool prof_on(int fam);
void prof_push(const ProfRec &r);

// Per-launch log of the dense convolutions (lsn_prof_launch_log): what was asked (the arguments of the entry point) and how long
// it took on the launch stream -- tools/instep_vs_isolated.py replays every logged call on its own and puts the two durations
// side by side.
struct LaunchRec {
    hipEvent_t e0, e1;
    int v[13];   // kind, C, Co, kh, kw, stride, pad, dil, relu, xpitch, nlv, has_res, has_gate
    int B[16], H[16], W[16];
};
bool launch_log_on();
void launch_log_push(const LaunchRec &r);

// RAII: events around the launches issued while the object lives
struct ProfSpan {
    ProfRec r;
    hipStream_t st;
    bool on;
    ProfSpan(int fam, double flops, double bytes, hipStream_t s) : st(s), on(prof_on(fam))
    {
        if (!on) return;
        r.fam = fam, r.flops = flops, r.bytes = bytes;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(r.e0, st);
    }
    ~ProfSpan()
    {
        if (!on) return;
        (void)hipEventRecord(r.e1, st);
        prof_push(r);
    }
    ProfSpan(const ProfSpan &) = delete;
    ProfSpan &operator=(const ProfSpan &) = delete;
};

}  // namespace lsn
