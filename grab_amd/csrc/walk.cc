// walk.cc -- see walk.h.
#include "walk.h"

#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct DirList {
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::string> dirs;
    size_t busy = 0; // walkers inside a directory: they may still add to `dirs`
    std::atomic<size_t> files{0};
};

void walk_thread(DirList &L, const std::function<void(std::string &&, const struct stat &)> &on_file)
{
    std::vector<std::string> sub;
    for (;;) {
        std::string dir;
        {
            std::unique_lock<std::mutex> lk(L.m);
            L.cv.wait(lk, [&] { return !L.dirs.empty() || L.busy == 0; });
            if (L.dirs.empty()) return; // nobody is inside a directory any more: the walk is over
            dir = std::move(L.dirs.back()); // depth first: keeps the list short
            L.dirs.pop_back();
            L.busy++;
        }
        sub.clear();
        if (DIR *d = opendir(dir.c_str())) { // (an unreadable directory is skipped: nftw's FTW_DNR, which the reference ignores)
            const int dfd = dirfd(d);
            const std::string prefix = dir == "/" ? dir : dir + "/";
            while (struct dirent *e = readdir(d)) {
                const char *n = e->d_name;
                if (n[0] == '.' && (n[1] == 0 || (n[1] == '.' && n[2] == 0))) continue;
                if (e->d_type == DT_LNK) continue; // FTW_PHYS: reported as FTW_SL, which the reference skips
                if (e->d_type == DT_DIR) {
                    sub.push_back(prefix + n);
                    continue;
                }
                if (e->d_type != DT_REG && e->d_type != DT_UNKNOWN) continue; // sockets, fifos, devices
                struct stat st;
                if (fstatat(dfd, n, &st, AT_SYMLINK_NOFOLLOW) != 0) continue;
                if (S_ISDIR(st.st_mode)) {
                    sub.push_back(prefix + n);
                } else if (S_ISREG(st.st_mode)) {
                    L.files++;
                    on_file(prefix + n, st);
                }
            }
            closedir(d);
        }
        {
            std::lock_guard<std::mutex> lk(L.m);
            for (std::string &s : sub) L.dirs.push_back(std::move(s));
            L.busy--;
            if (!sub.empty() || L.busy == 0) L.cv.notify_all();
        }
    }
}

} // namespace

size_t grab_walk(const std::string &root_in, int threads, const std::function<void(std::string &&, const struct stat &)> &on_file)
{
    std::string root = root_in;
    while (root.size() > 1 && root.back() == '/') root.pop_back(); // as nftw does
    struct stat st;
    if (lstat(root.c_str(), &st) != 0) return 0;
    if (S_ISREG(st.st_mode)) {
        on_file(std::move(root), st);
        return 1;
    }
    if (!S_ISDIR(st.st_mode)) return 0; // a symbolic link as the root is FTW_SL under FTW_PHYS: not followed
    DirList L;
    L.dirs.push_back(root);
    threads = std::max(1, std::min(threads, 64));
    std::vector<std::thread> pool;
    for (int i = 1; i < threads; i++) pool.emplace_back([&] { walk_thread(L, on_file); });
    walk_thread(L, on_file);
    for (std::thread &t : pool) t.join();
    return L.files.load();
}
