// tsd/fr.hpp -- the French keyword spellings libtsd's call sites are written with (core/include/tsd/fr.hpp:5-36):
// plain aliases of C++ keywords, so that existing libtsd code (README examples, tests) compiles unchanged
// against the mirror.
#pragma once
#ifndef soit
#define let auto
#define soit auto
#define Soient auto
#define soient auto
#define Si if
#define si if
#define sinon else
#define retourne return
#define Pour for
#define pour for
#define Tantque while
#define tantque while
#define ou ||
#define et &&
#endif
namespace tsd {
static const bool non = false, oui = true;
}
using tsd::non;
using tsd::oui;
