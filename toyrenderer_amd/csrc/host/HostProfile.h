// HostProfile.h -- TRHOST_PROFILE=1: accumulated host time of named scopes, printed at process exit.
// Diagnostics for the per-frame recording cost of the host mirror; a disabled scope is one branch.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace hostprofile
{
struct Table
{
    bool on = getenv("TRHOST_PROFILE") != nullptr;
    std::mutex m;
    std::map<std::string, std::pair<unsigned long long, double>> acc;
    ~Table()
    {
        if (!on) return;
        for (auto& kv : acc)
            fprintf(stderr, "[trhost profile] %-40s calls %8llu  total %10.1f us  avg %7.2f us\n", kv.first.c_str(), kv.second.first,
                    kv.second.second, kv.second.second / (double)kv.second.first);
    }
};
inline Table& table() { static Table t; return t; }

struct Scope
{
    const char* name;
    std::chrono::steady_clock::time_point t0;
    explicit Scope(const char* n) : name(table().on ? n : nullptr) { if (name) t0 = std::chrono::steady_clock::now(); }
    ~Scope()
    {
        if (!name) return;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        Table& t = table();
        std::lock_guard<std::mutex> lk(t.m);
        auto& e = t.acc[name];
        e.first++; e.second += us;
    }
};
}
#define HOST_PROFILE_CAT2(a, b) a##b
#define HOST_PROFILE_CAT(a, b) HOST_PROFILE_CAT2(a, b)
#define HOST_PROFILE_SCOPE(name) ::hostprofile::Scope HOST_PROFILE_CAT(hostProfileScope, __LINE__)(name)
