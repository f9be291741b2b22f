// Host-side C++ mirror of the reference's Scala-facing interface for the hot path,
// layered on the C ABI (include/rt_dp.h).  Same names, same argument meaning, same
// error behaviour as the Chisel module and the chiseltest driver calls it replaces:
//
//   RaytracerParams                      UnifiedDatapath.scala:13-23
//   UnifiedDatapathOpCode                IOSpec.scala:10-24
//   EnhancedInputBundle / OutputBundle   IOSpec.scala:106-173
//   UnifiedDatapath(p)                   UnifiedDatapath.scala:33-59 (elaboration asserts :36-47)
//   in.enqueue / in.enqueueSeq           chiseltest DecoupledDriver, as used in
//   out.waitForValid / out.bits / step   src/test/scala/raytracer_datapath/Datapath_test.scala:476-496
//
// (paths relative to the reference's src/main/scala/raytracer_datapath/ unless longer).
// The RTL moves one record per clock; here records queue up on `in` and are evaluated
// on the GPU in one batch the first time `out` is looked at.  Results come back in
// order, one per input record, like the RTL's in-order pipe.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_dp.h"

namespace raytracer_datapath {

struct RaytracerParams {
    bool io_recorded_float = false;
    bool internal_recorded_float = true;
    std::optional<int> support_euclidean = std::nullopt;  // None, or Some(16)
};

enum class UnifiedDatapathOpCode : uint32_t { OpTriangle = 0, OpQuadbox = 1, OpEuclidean = 2, OpAngular = 3 };

using EnhancedInputBundle = rt_dp_input_bundle;
using EnhancedOutputBundle = rt_dp_output_bundle;

inline uint32_t floatToBits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bitsToFloat(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

class UnifiedDatapath {
  public:
    // Elaboration: an illegal parameter set throws with the RTL's assertion text.
    explicit UnifiedDatapath(const RaytracerParams &p = RaytracerParams(), int device = -1) : in(*this), out(*this) {
        rt_dp_config cfg{};
        cfg.struct_size = sizeof cfg;
        cfg.device = device;
        cfg.io_recorded_float = p.io_recorded_float ? 1u : 0u;
        cfg.internal_recorded_float = p.internal_recorded_float ? 1u : 0u;
        cfg.support_euclidean = p.support_euclidean ? (uint32_t)*p.support_euclidean : 0u;
        const int rc = rt_dp_create(&cfg, &ctx_);
        if (rc == RT_DP_EINVAL) throw std::invalid_argument(rt_dp_last_error());
        if (rc != RT_DP_OK) throw std::runtime_error(rt_dp_last_error());
    }
    ~UnifiedDatapath() { rt_dp_destroy(ctx_); }
    UnifiedDatapath(const UnifiedDatapath &) = delete;
    UnifiedDatapath &operator=(const UnifiedDatapath &) = delete;

    // `in`: Flipped(Decoupled(EnhancedInputBundle))
    struct InPort {
        explicit InPort(UnifiedDatapath &d) : d_(d) {}
        bool ready() const { return true; }  // a batch evaluator never back-pressures
        void enqueue(const EnhancedInputBundle &b) { d_.pending_.push_back(b); }
        void enqueueSeq(const std::vector<EnhancedInputBundle> &s) { d_.pending_.insert(d_.pending_.end(), s.begin(), s.end()); }
      private:
        UnifiedDatapath &d_;
    } in;

    // `out`: Decoupled(EnhancedOutputBundle)
    struct OutPort {
        explicit OutPort(UnifiedDatapath &d) : d_(d) {}
        bool valid() { d_.flush(); return !d_.done_.empty(); }
        void waitForValid() { if (!valid()) throw std::runtime_error("waitForValid: no record in flight"); }
        const EnhancedOutputBundle &bits() { waitForValid(); return d_.done_.front(); }  // peek
        void step() { waitForValid(); d_.done_.pop_front(); }                               // clock.step() with ready high
        EnhancedOutputBundle dequeue() { EnhancedOutputBundle b = bits(); d_.done_.pop_front(); return b; }
      private:
        UnifiedDatapath &d_;
    } out;

    // Evaluate everything queued on `in` (one GPU batch).
    void flush() {
        if (pending_.empty()) return;
        std::vector<EnhancedOutputBundle> res(pending_.size());
        const int rc = rt_dp_eval_bundles(ctx_, pending_.data(), res.data(), pending_.size());
        if (rc != RT_DP_OK) throw std::runtime_error(rt_dp_last_error());
        done_.insert(done_.end(), res.begin(), res.end());
        pending_.clear();
    }

    // The RTL's reset: accumulators to +0; queued records are dropped like in-flight pipeline state.
    void reset() {
        pending_.clear();
        done_.clear();
        if (rt_dp_reset(ctx_) != RT_DP_OK) throw std::runtime_error(rt_dp_last_error());
    }

    rt_dp_ctx *context() { return ctx_; }

  private:
    rt_dp_ctx *ctx_ = nullptr;
    std::vector<EnhancedInputBundle> pending_;
    std::deque<EnhancedOutputBundle> done_;
};

}  // namespace raytracer_datapath
