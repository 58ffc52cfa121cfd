// walk.h -- the tree walk of `grab -n`: several threads, no barrier between walking and scanning.
//
// The reference walks the whole tree on one thread (nftw, /root/reference/src/main.cc:178) and only then starts
// its worker threads (main.cc:195-216); its README describes a "lockfree parallel nftw()" on the other branch
// (README.md:137-139).  Here directories are units of work: a walker takes one, reads it, hands
// every regular file to the consumer and puts the sub-directories back on the shared list for whoever is free --
// the walk spreads over its threads by itself, whatever the shape of the tree, and files reach the scan queue
// while the walk is still going on.
//
// What is reported is what nftw(path, fn, 1024, FTW_PHYS) reports to the reference's callback as
// FTW_F && S_ISREG (grab.cc:265-266, main.cc:76-77): regular files only, symbolic links neither followed nor
// reported (a symbolic link given as the root included), unreadable directories skipped, path strings built the
// way nftw builds them (trailing slashes of the root dropped, "dir/name").  Order is unspecified.
#pragma once

#include <sys/stat.h>

#include <functional>
#include <string>

// `on_file` is called concurrently from the walker threads.  Returns the number of files reported.
size_t grab_walk(const std::string &root, int threads, const std::function<void(std::string &&path, const struct stat &st)> &on_file);
