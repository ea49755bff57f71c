// dsp/dsp-all.hpp -- the English umbrella header (core/include/dsp/dsp-all.hpp), hot-path subset.
#pragma once
#include "tsd/fr.hpp"
#include "dsp/dsp.hpp"
#include "dsp/filter.hpp"
#include "dsp/fourier.hpp"
using namespace dsp;
using namespace dsp::filter;
using namespace dsp::fourier;
