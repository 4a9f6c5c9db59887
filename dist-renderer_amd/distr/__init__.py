"""distr -- MI355X-native differentiable sphere tracing (host side of the C ABI in include/distr.h)."""
