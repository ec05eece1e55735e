/* Forced-include shim used ONLY when compiling the unmodified reference sources
 * (see Makefile.ref).  The reference calls an unqualified isnan()
 * (src/Simulation.cpp:2716), which GCC >= 6 no longer exposes in the global
 * namespace from <cmath>.  This is a toolchain-compat flag, not a stand-in
 * for any reference header. */
#ifdef __cplusplus
#include <cmath>
using std::isnan;
#endif
