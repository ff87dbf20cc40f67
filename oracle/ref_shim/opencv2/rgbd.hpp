/* oracle/_ref shim: see opencv2/core/core.hpp */
#include "opencv2/core/core.hpp"
