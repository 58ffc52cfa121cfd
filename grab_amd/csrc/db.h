// db.h -- the opaque gscan_db of include/gscan.h: a compiled pattern (pattern.h), shared by engine.hip and matcher.cc.
#pragma once
#include "pattern.h"

struct gscan_db {
    gscan::Database db;
};
