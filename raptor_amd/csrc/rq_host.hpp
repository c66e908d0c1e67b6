// rq_host.hpp — what the host-side translation units of libraptor_quad.so share (rq_capi*.cpp, rq_comm.cpp):
// error reporting, the status-returning check macros and the current-device scope.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>

#include "../../include/raptor_quad.h"

namespace rq {

// records the message rq_last_error() returns on this thread; returns `status`
int fail(int status, const std::string& msg);

extern thread_local int tl_scope_depth, tl_scope_device;

// Every entry point runs on its rq_device's HIP device and leaves the calling thread's current device as it
// found it: a host that drives several GPUs from one thread (or PyTorch with another current device) must
// not find its device switched behind its back.
struct DeviceScope {
    int previous = -1, target, rc = RQ_OK;
    bool nested = false;                 // inside another scope of the same device: nothing to query or restore
    explicit DeviceScope(int ordinal) : target(ordinal) {
        if (tl_scope_depth > 0 && tl_scope_device == target) { nested = true; ++tl_scope_depth; return; }
        if (hipGetDevice(&previous) != hipSuccess) previous = -1;
        if (previous != target) {
            const hipError_t e = hipSetDevice(target);
            if (e != hipSuccess) rc = fail(RQ_ERR_HIP, std::string("hipSetDevice -> ") + hipGetErrorString(e));
        }
        outer_depth = tl_scope_depth; outer_device = tl_scope_device;
        tl_scope_depth = 1; tl_scope_device = target;
    }
    explicit DeviceScope(const rq_device* dev);
    DeviceScope(const rq_device* dev, struct KeepResident);
    ~DeviceScope() {
        if (nested) { --tl_scope_depth; return; }
        tl_scope_depth = outer_depth; tl_scope_device = outer_device;
        if (previous >= 0 && previous != target) (void)hipSetDevice(previous);
    }
    int outer_depth = 0, outer_device = -1;
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// read-only views of the opaque objects for the other host translation units
int device_ordinal(const rq_device* dev);
hipStream_t device_stream(const rq_device* dev);
rq_device* env_device(const rq_env* env);
uint32_t env_num_envs(const rq_env* env);
const float* env_finished_returns(const rq_env* env);      // device [ld]

// The resident executor of the small-batch loop (rq_capi_vector.cpp resident_*) works outside the device's stream; any entry point that may touch
// that stream first retires it (a few microseconds, and only when one is running): that is this hook, run by every DeviceScope made
// from an rq_device.  The three calls of the loop itself construct their scope with KeepResident and retire explicitly on their slow paths.
int resident_scope_hook(const rq_device* dev);
struct KeepResident {};

inline DeviceScope::DeviceScope(const rq_device* dev) : DeviceScope(device_ordinal(dev)) {
    if (rc == RQ_OK) rc = resident_scope_hook(dev);
}
inline DeviceScope::DeviceScope(const rq_device* dev, KeepResident) : DeviceScope(device_ordinal(dev)) {}

}  // namespace rq

#define RQ_REQUIRE(cond, status, msg)                                                 \
    do {                                                                              \
        if (!(cond)) return rq::fail((status), std::string(__func__) + ": " + (msg)); \
    } while (0)

#define RQ_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return rq::fail(e_ == hipErrorOutOfMemory ? RQ_ERR_OUT_OF_MEMORY : RQ_ERR_HIP,        \
                            std::string(__func__) + ": " #expr " -> " + hipGetErrorString(e_));   \
    } while (0)
