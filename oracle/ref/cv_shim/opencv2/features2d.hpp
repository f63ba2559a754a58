// (see cvshim.hpp)
#include "cvshim.hpp"
