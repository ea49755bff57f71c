// tsd/tsd-all.hpp -- everything of the mirror at once, namespaces opened like libtsd's own umbrella header
// (core/include/tsd/tsd-all.hpp): the hot-path subset only (no vue / telecom / stats here).
#pragma once
#include "tsd/fr.hpp"
#include "tsd/tsd.hpp"
#include "tsd/filtrage.hpp"
#include "tsd/fourier.hpp"
using namespace tsd;
using namespace tsd::filtrage;
using namespace tsd::fourier;
