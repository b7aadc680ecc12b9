// ORACLE — TEST INFRASTRUCTURE ONLY (see pgo_common.h header).
//
// Room_Generator shared by caveflyer and jumper (games/caveflyer/room_generator.{h,cpp}; jumper's copy is identical):
// cellular automaton, 4-connected rooms collected in real std::unordered_set<int> (their iteration order reaches
// the level through `for (int i : best_room)`), BFS path, 8-neighbour widening.
#pragma once

#include <queue>
#include <unordered_set>
#include <vector>

namespace pgo {

struct Rooms {
    int gw = 0, gh = 0;
    std::vector<int> grid;
    int get(int x, int y) const {
        if (x < 0 || y < 0 || x >= gw || y >= gh) return 1;
        return grid[y + gh * x];
    }
    void update() {  // :20-35
        std::vector<int> nxt(grid.size());
        for (int i = 0; i < static_cast<int>(grid.size()); i++) {
            const int x = i / gh, y = i % gh;
            int n = 0;
            for (int a = -1; a <= 1; a++)
                for (int b = -1; b <= 1; b++)
                    if (get(x + a, y + b) == 1) n++;
            nxt[i] = n >= 5 ? 1 : 0;
        }
        grid = nxt;
    }
    void build_room(int index, std::unordered_set<int>& room) const {  // :37-75
        std::queue<int> q;
        if (grid[index] != 0) return;
        q.push(index);
        while (!q.empty()) {
            const int cur = q.front();
            q.pop();
            if (grid[cur] != 0) continue;
            const int x = cur / gh, y = cur % gh;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++)
                    if ((i == 0 || j == 0) && (i + j != 0)) {
                        const int nx = x + i, ny = y + j;
                        if (nx < 0 || ny < 0 || nx >= gw || ny >= gh) continue;
                        const int ni = ny + gh * nx;
                        if (room.find(ni) == room.end() && grid[ni] == 0) {
                            q.push(ni);
                            room.insert(ni);
                        }
                    }
        }
    }
    void find_best_room(std::unordered_set<int>& best) const {  // :138-160
        std::unordered_set<int> all;
        best.clear();
        int best_size = -1;
        for (int i = 0; i < static_cast<int>(grid.size()); i++)
            if (grid[i] == 0 && all.find(i) == all.end()) {
                std::unordered_set<int> room;
                build_room(i, room);
                all.insert(room.begin(), room.end());
                if (static_cast<int>(room.size()) > best_size) {
                    best_size = static_cast<int>(room.size());
                    best = room;
                }
            }
    }
    void find_path(int src, int dst, std::vector<int>& path) const {  // :77-136
        std::unordered_set<int> covered;
        if (grid[src] != 0) return;
        std::vector<int> expanded{src}, parents{-1};
        int at = 0;
        while (at < static_cast<int>(expanded.size())) {
            const int cur = expanded[at];
            if (cur == dst) break;
            const int x = cur / gh, y = cur % gh;
            for (int i = -1; i <= 1; i++)
                for (int j = -1; j <= 1; j++)
                    if ((i == 0 || j == 0) && (i + j != 0)) {
                        const int nx = x + i, ny = y + j;
                        if (nx < 0 || ny < 0 || nx >= gw || ny >= gh) continue;
                        const int ni = ny + gh * nx;
                        if (covered.find(ni) == covered.end() && grid[ni] == 0) {
                            expanded.push_back(ni);
                            parents.push_back(at);
                            covered.insert(ni);
                        }
                    }
            at++;
        }
        if (at < static_cast<int>(expanded.size()) && expanded[at] == dst) {
            std::vector<int> tmp;
            while (at >= 0) {
                tmp.push_back(expanded[at]);
                at = parents[at];
            }
            path.assign(tmp.rbegin(), tmp.rend());
        }
    }
    void expand_room(std::unordered_set<int>& set, int n) const {  // :162-202
        std::unordered_set<int> cur;
        cur.insert(set.begin(), set.end());
        for (int loop = 0; loop < n; loop++) {
            std::unordered_set<int> nxt;
            for (int c : cur) {
                if (grid[c] != 0) continue;
                const int x = c / gh, y = c % gh;
                for (int i = -1; i <= 1; i++)
                    for (int j = -1; j <= 1; j++)
                        if (i != 0 || j != 0) {
                            const int nx = x + i, ny = y + j;
                            if (nx < 0 || ny < 0 || nx >= gw || ny >= gh) continue;
                            const int ni = ny + gh * nx;
                            if (set.find(ni) == set.end() && grid[ni] == 0) {
                                set.insert(ni);
                                nxt.insert(ni);
                            }
                        }
            }
            cur = nxt;
        }
    }
};

}  // namespace pgo
