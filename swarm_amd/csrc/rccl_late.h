// rccl_late.h — RCCL bound at run time, by the first multi-rank job only.
// librccl.so is 570 MB of code objects for every collective RCCL has; a single-GPU run — the usual one — needs none
// of it, and as a link-time dependency of libswarm_amd.so it was mapped, relocated and registered with the HIP
// runtime at every start of `swarm`.  swa_multi_create (multi.hip) asks for it when ranks on distinct devices exist.
#pragma once

#include <dlfcn.h>

#include <rccl/rccl.h>        // types and prototypes only: no symbol of it is linked

#include <mutex>
#include <string>

struct swa_rccl_api {
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  std::string error;           // why the library could not be bound (empty: bound)
};

inline const swa_rccl_api & swa_rccl() {
  static swa_rccl_api api;
  static std::once_flag once;
  std::call_once(once, [] {
    void * lib = nullptr;
    for (const char * name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (lib != nullptr) { break; }
    }
    if (lib == nullptr) { const char * e = dlerror(); api.error = std::string("librccl.so: ") + (e != nullptr ? e : "not found"); return; }
    auto bind = [&](auto & fn, const char * symbol) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(lib, symbol));
      if (fn == nullptr && api.error.empty()) { api.error = std::string("librccl.so lacks ") + symbol; }
    };
    bind(api.CommInitAll, "ncclCommInitAll");
    bind(api.CommDestroy, "ncclCommDestroy");
    bind(api.GetErrorString, "ncclGetErrorString");
    bind(api.GroupStart, "ncclGroupStart");
    bind(api.GroupEnd, "ncclGroupEnd");
    bind(api.Send, "ncclSend");
    bind(api.Recv, "ncclRecv");
    bind(api.Broadcast, "ncclBroadcast");
    bind(api.AllReduce, "ncclAllReduce");
  });
  return api;
}
