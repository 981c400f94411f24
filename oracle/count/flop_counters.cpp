// oracle/count/flop_counters.cpp -- TEST INFRASTRUCTURE: storage and C access of the per-thread operation counters
// of the counting build of the oracle (flop_count.hpp).
#include "flop_count.hpp"
#undef double
thread_local OracleFlopCounters oracle_flops = {0, 0, 0, 0, 0};
extern "C" void oracle_flops_reset(void) { oracle_flops = OracleFlopCounters{0, 0, 0, 0, 0}; }
extern "C" void oracle_flops_read(unsigned long long *out5)
{
    out5[0] = oracle_flops.add; out5[1] = oracle_flops.mul; out5[2] = oracle_flops.div;
    out5[3] = oracle_flops.sqrt_; out5[4] = oracle_flops.trans;
}
