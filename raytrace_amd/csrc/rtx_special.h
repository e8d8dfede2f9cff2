// rtx_special.h — kernels compiled for one scene at scene upload (host side).
//
// The fast-mode linear scan walks a table of records with wave-uniform scalar loads; every record
// field is then a scalar-register operand, which halves the issue rate of the vector instruction
// that reads it, and every record costs a load, a wait and loop bookkeeping.  For a scene of a few
// dozen primitives the table can be compiled into the kernel instead: rtx_upload_scene writes the
// scan out as straight-line code with the records as literals and compiles the trace kernel around
// it with the HIP run-time compiler.  The reference has nothing comparable (its scene is a tree of
// virtual calls, ShapeList.cpp:5-17); this is the counterpart of building the scene into the binary.
#pragma once
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "rtx.h"
#include "rtx_device.h"

namespace rtxspecial {

// Which trace kernel a scene's frames run (what rtx_fast_mode::launch_trace picks for a scan scene):
// the one that walks a light tree, the one for a light list with a constant background, or the plain one.
struct Variant {
    bool light_tree = false, list_lit = false;
};

// The kernels compiled for one scene, loaded on one device: the fast-mode trace kernel for frames
// (trace_kernel<false, false, light_tree, list_lit>), the one with per-sample output
// (trace_kernel<false, true, ...>) and the ray query (hit_kernel<false>), so that a scene's
// frames, samples and queries come from one and the same scan.
struct Kernel {
    hipModule_t module = nullptr;
    hipFunction_t frames = nullptr, samples = nullptr, hit = nullptr;
};

// Primitive counts above which the scan is not specialised (the straight-line code of a long table
// no longer fits the instruction cache; such scenes are rendered through the tree anyway).
constexpr size_t kMaxRecords = 64;

// The scan of the bucketed table (groups / recs as build_scan_table lays them out) as the definition
// of the RTX_SPECIAL_SCAN macro.
std::string scan_source(const std::vector<rtxd::ScanGroup>& groups, const std::vector<rtxd::ScanRec>& recs);

// The top-level entries of a light list without a tree (or the one bare light shape) as the
// definition of RTX_SPECIAL_LIGHTS / RTX_SPECIAL_LIGHT_WEIGHT: the density sum of a diffuse bounce
// (ShapeList.cpp:19-26) then runs over literals instead of scalar-loaded records.  Empty for scenes
// without lights or with more than kMaxRecords of them.  A kernel's text is scan_source + lights_source.
std::string lights_source(const std::vector<rtxd::Light>& lights, uint32_t n_lights, bool is_list);

// What the scene holds (textured surfaces, the material kinds of its hit records, the kinds of its
// light entries) as the definition of RTX_SPECIAL_FEATURES: tests for what it does not hold are
// decided when its kernels are compiled.  `hits` are the records of the scan table's positions.
std::string features_source(const std::vector<rtxd::ScanHit>& hits, const std::vector<rtxd::Light>& lights, uint32_t n_lights,
                            bool light_tree);

// Compiles the kernels around `scan` for gfx950 and loads them on the current device.  Code objects are kept
// per process, variant and scan text, so a scene uploaded again - or on another device - is not compiled
// twice.  Returns false with the reason in `why` (no run-time compiler in the process, compilation
// failed, ...): the caller then keeps the table-driven kernel.
bool load(const std::string& text, Variant variant, Kernel& out, std::string& why, double* compile_seconds);

// The compilation alone (no device needed): fills the process's cache so that a later load of the
// same scan only loads the code object.
bool prepare(const std::string& text, Variant variant, std::string& why, double* compile_seconds);

void unload(Kernel& k);

// A compilation in the background (RTX_OPTION_SPECIALIZE_SCAN = 2): a thread of the library runs the
// helper program rtx_compile (beside librtx.so) on the text, waits for it and puts the code object
// into the process's cache; the owner polls `state` and, once it is not 0, loads the code object with
// load().  The compiler itself never runs on a background thread of this process: a process may
// exit at any time, and a compiler caught by the teardown of its own static objects aborts, faults or
// hangs - a helper that is left behind just ends.  finish() ends a compilation nobody waits for any
// more (the helper is terminated) and joins the thread; a job whose owner never calls it is harmless.
struct Job {
    std::string text;
    Variant variant;
    std::atomic<int> state{0};  // 0 compiling, 1 in the cache, 2 failed
    std::string why;            // (written before state becomes 2)
    double seconds = 0.0;
    std::atomic<int> helper_pid{0};
    std::thread thread;
    // (the thread shares the job with its owner: whoever drops it last - the thread itself, if the owner never
    // called finish() - must not destroy a thread that can still be joined)
    ~Job() {
        if (thread.joinable()) thread.detach();
    }
};
std::shared_ptr<Job> start(const std::string& text, Variant variant);
void finish(const std::shared_ptr<Job>& job);

// The helper's side: compiles and writes the answer of the wire format (rtx_special.cpp) into `reply`.
bool compile_for_helper(const std::string& text, Variant variant, std::string& reply);

}  // namespace rtxspecial
