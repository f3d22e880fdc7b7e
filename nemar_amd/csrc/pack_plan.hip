// nemar_amd — weight-pack plans: see pack_plan.h.  C ABI (include/nemar_hip.h):
//   nemar_pack_plan_record(plan)   plan >= 0: pack launches issued by the convolution entry points on THIS thread are also recorded
//                                  into `plan` (and still executed); -1: stop
//   nemar_pack_plan_jobs / _bytes  jobs recorded so far / bytes of device memory their arguments need
//   nemar_pack_plan_commit         copy the job arguments into a caller-owned device buffer (once per change of the plan)
//   nemar_pack_plan_run            re-run every recorded job: <= 6 launches
//   nemar_pack_plan_reset          forget the plan
// The caller owns the device buffer and keeps it (and every weight / packed-image buffer the jobs point at) alive while the plan is
// in use.  A plan belongs to one optimizer: its jobs are the packs of that optimizer's weights between two of its steps.
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "pack_plan.h"

namespace {

struct Family {
    size_t args_bytes = 0;
    nemar_pack_multi_fn fn = nullptr;
};
Family g_fam[PACK_FAMS];

struct Plan {
    std::vector<unsigned char> args[PACK_FAMS];     // host copies of the job arguments, per family
    int njobs[PACK_FAMS] = {0, 0, 0, 0, 0, 0, 0};
    int gx[PACK_FAMS] = {0, 0, 0, 0, 0, 0, 0}, gy[PACK_FAMS] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned char* dev = nullptr;             // committed image: the families' argument arrays back to back (256-byte aligned)
    size_t dev_off[PACK_FAMS] = {0, 0, 0, 0, 0, 0, 0};
    bool dirty = true;
};
std::map<int, Plan> g_plans;
std::mutex g_mu;
thread_local int t_recording = -1;

size_t plan_bytes(Plan& p) {
    size_t o = 0;
    for (int f = 0; f < PACK_FAMS; ++f) {
        p.dev_off[f] = o;
        o += (p.args[f].size() + 255) & ~(size_t)255;
    }
    return o ? o : 256;
}

__device__ __forceinline__ void max_parts_body(const NemarPackMaxArgs& a, int bx, int, int gx) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    for (long long i = (long long)bx * 256 + threadIdx.x; i < a.n; i += 256ll * gx) {
        const unsigned u = __builtin_bit_cast(unsigned, a.x[i]) & 0x7fffffffu;
        m = max(m, u < 0x7f800000u ? u : 0u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) a.out[bx] = max(max(red[0], red[1]), max(red[2], red[3]));
}
NEMAR_PACK_MULTI(max_parts_multi_kernel, NemarPackMaxArgs, max_parts_body, 256)
void max_parts_multi(const void* jobs, int njobs, int gx, int gy, hipStream_t st) {
    hipLaunchKernelGGL(max_parts_multi_kernel, dim3(gx, gy, njobs), dim3(256), 0, st, (const NemarPackMaxArgs*)jobs);
}
struct RegMax {
    RegMax() { nemar_pack_register(PACK_FAM_MAX, sizeof(NemarPackMaxArgs), max_parts_multi); }
} g_reg_max;

}  // namespace

void nemar_pack_register(int fam, size_t args_bytes, nemar_pack_multi_fn fn) {
    g_fam[fam].args_bytes = args_bytes;
    g_fam[fam].fn = fn;
}

bool nemar_pack_recording() { return t_recording >= 0; }

void nemar_pack_record_job(int fam, const void* args, int gx, int gy) {
    if (t_recording < 0 || !g_fam[fam].fn) return;
    std::lock_guard<std::mutex> lk(g_mu);
    Plan& p = g_plans[t_recording];
    const unsigned char* a = (const unsigned char*)args;
    p.args[fam].insert(p.args[fam].end(), a, a + g_fam[fam].args_bytes);
    ++p.njobs[fam];
    if (gx > p.gx[fam]) p.gx[fam] = gx;
    if (gy > p.gy[fam]) p.gy[fam] = gy;
    p.dirty = true;
}

NEMAR_API int nemar_pack_plan_record(int plan) {
    t_recording = plan;
    return NEMAR_OK;
}

NEMAR_API int nemar_pack_plan_jobs(int plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(plan);
    if (it == g_plans.end()) return 0;
    int n = 0;
    for (int f = PACK_FAM_EXACT; f < PACK_FAMS; ++f) n += it->second.njobs[f];      // (packs; the max jobs ride along)
    return n;
}

NEMAR_API size_t nemar_pack_plan_bytes(int plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(plan);
    return it == g_plans.end() ? 0 : plan_bytes(it->second);
}

NEMAR_API int nemar_pack_plan_commit(int plan, void* device_buffer, size_t bytes, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(plan);
    NEMAR_REQUIRE(it != g_plans.end() && device_buffer, "pack_plan_commit: unknown plan / null buffer");
    Plan& p = it->second;
    const size_t need = plan_bytes(p);
    if (bytes < need) {
        nemar_set_error("pack_plan_commit: buffer %zu < %zu", bytes, need);
        return NEMAR_EWORKSPACE;
    }
    for (int f = 0; f < PACK_FAMS; ++f)
        if (!p.args[f].empty())
            NEMAR_HIP_CALL(hipMemcpyAsync((unsigned char*)device_buffer + p.dev_off[f], p.args[f].data(), p.args[f].size(), hipMemcpyHostToDevice,
                                          (hipStream_t)stream));
    // (the sources are this plan's own host vectors, pageable: wait, so that a job recorded right afterwards cannot move them under the copy)
    NEMAR_HIP_CALL(hipStreamSynchronize((hipStream_t)stream));
    p.dev = (const unsigned char*)device_buffer;
    p.dirty = false;
    return NEMAR_OK;
}

NEMAR_API int nemar_pack_plan_dirty(int plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(plan);
    return it == g_plans.end() ? 0 : (it->second.dirty ? 1 : 0);
}

NEMAR_API int nemar_pack_plan_run(int plan, void* stream) {
    NEMAR_CLEAR_HIP_ERROR();
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(plan);
    NEMAR_REQUIRE(it != g_plans.end(), "pack_plan_run: unknown plan %d", plan);
    Plan& p = it->second;
    NEMAR_REQUIRE(!p.dirty && p.dev, "pack_plan_run: plan %d has uncommitted jobs", plan);
    for (int f = 0; f < PACK_FAMS; ++f)                      // (stage order = family order: the max reductions first)
        if (p.njobs[f]) g_fam[f].fn(p.dev + p.dev_off[f], p.njobs[f], p.gx[f], p.gy[f], (hipStream_t)stream);
    NEMAR_CHECK_LAUNCH("pack_plan_run");
    return NEMAR_OK;
}

NEMAR_API int nemar_pack_plan_reset(int plan) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_plans.erase(plan);
    return NEMAR_OK;
}
