// include/compat/include/misc_image.h -- stands where the reference's include/misc_image.h stands (/root/reference/include/misc_image.h): an application that includes the reference's headers by their
// relative paths (synthetic-hand-tracker/synthetic-tracker.cpp:15-24) compiles its tracker side against the MI355X binding unchanged when this directory's include/ and
// third_party/ take the place of the reference's.  Every one of these files forwards to the same header, include/ht_formats.hpp (which pulls in include/ht_handtrack.hpp):
// the reference's class and function names at global scope over the C-ABI (include/ht_mi355x.h).  The application's own LoadAnimBank (synthetic-tracker.cpp:39-55) does not
// collide: the binding's lives in namespace ht_mi355x (its Pose extraction operator is found through the argument's namespace).
// Not covered: the window side (glwin.h, misc_gl.h and the 4x4 matrix helpers they use) stays the reference's own code (out of scope).
#pragma once
#ifndef HT_MI355X_GLOBAL_NAMES
#define HT_MI355X_GLOBAL_NAMES
#endif
#include "../../ht_formats.hpp"
