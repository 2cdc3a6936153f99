// Force-included (-include) into the reference sampler TUs: replaces
// std::random_device with a settable, deterministic source so the verbatim
// reference samplers become reproducible (single-threaded use only).
#ifndef GLX_ORACLE_FIXED_RD_H_
#define GLX_ORACLE_FIXED_RD_H_
#include <random>
namespace std {
struct glx_fixed_random_device {
  typedef unsigned int result_type;
  static unsigned int& seed() { static unsigned int s = 20240923u; return s; }
  result_type operator()() { return seed(); }
  static constexpr result_type min() { return 0; }
  static constexpr result_type max() { return 0xffffffffu; }
  double entropy() const { return 0.0; }
};
}  // namespace std
#define random_device glx_fixed_random_device
#endif
