// Library-level entry points: version, status strings, thread-local error text.
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "wmd_internal.h"

namespace wmd {

static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

bool g_prof_on = false;
namespace {
struct ProfRec {
    std::string name;
    hipEvent_t e0, e1;
    double flops, bytes, mfma_flops;
};
std::vector<ProfRec> g_prof;
}  // namespace

int prof_open(const char* name, double flops, double bytes, hipStream_t s) {
    ProfRec r;
    r.name = name;
    r.flops = flops;
    r.bytes = bytes;
    r.mfma_flops = flops;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
    hipEventRecord(r.e0, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
}

void prof_close(int idx, hipStream_t s) { hipEventRecord(g_prof[idx].e1, s); }

void prof_set_mfma(int idx, double mfma_flops) { g_prof[idx].mfma_flops = mfma_flops; }

}  // namespace wmd

extern "C" int wmd_profile_begin(void) {
    for (auto& r : wmd::g_prof) {
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    wmd::g_prof.clear();
    wmd::g_prof_on = true;
    return WMD_OK;
}

extern "C" long wmd_profile_end(char* buf, size_t cap) {
    wmd::g_prof_on = false;
    struct Agg {
        long calls = 0;
        double ms = 0, flops = 0, bytes = 0, mfma = 0;
    };
    std::map<std::string, Agg> agg;
    std::vector<std::string> order;
    for (auto& r : wmd::g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess)
            return wmd::fail(WMD_ERR_HIP, "wmd_profile_end: event query failed for %s", r.name.c_str());
        if (!agg.count(r.name)) order.push_back(r.name);
        Agg& a = agg[r.name];
        a.calls++;
        a.ms += ms;
        a.flops += r.flops;
        a.bytes += r.bytes;
        a.mfma += r.mfma_flops;
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
    }
    wmd::g_prof.clear();
    std::string out = "[";
    for (size_t i = 0; i < order.size(); ++i) {
        const Agg& a = agg[order[i]];
        char line[512];
        snprintf(line, sizeof(line), "%s{\"kernel\": \"%s\", \"calls\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e, \"mfma_flops\": %.6e}",
                 i ? ", " : "", order[i].c_str(), a.calls, a.ms, a.flops, a.bytes, a.mfma);
        out += line;
    }
    out += "]";
    if (buf && cap) {
        size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}

extern "C" int wmd_version(void) { return WMD_VERSION; }

extern "C" const char* wmd_last_error(void) { return wmd::g_last_error; }

extern "C" const char* wmd_status_string(int status) {
    switch (status) {
        case WMD_OK: return "ok";
        case WMD_ERR_BAD_ARG: return "bad argument";
        case WMD_ERR_BAD_SHAPE: return "bad shape";
        case WMD_ERR_UNSUPPORTED: return "unsupported";
        case WMD_ERR_HIP: return "HIP error";
        case WMD_ERR_WORKSPACE: return "workspace too small";
        case WMD_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}
