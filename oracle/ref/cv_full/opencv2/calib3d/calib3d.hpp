// (oracle/ref/cv_full: every OpenCV header is the one stand-in)
#include "../../cvfull.hpp"
