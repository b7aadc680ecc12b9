#!/usr/bin/env python3
"""Copy the sprite/background PNGs a game needs from the reference tree into procgen2_amd/assets/.

Assets are *data* (MIT-licensed art shipped with the reference, see procgen2_amd/assets/NOTICE);
the GPU box only receives this repo, so the PNGs the engine loads have to live in it.  Run in
the build container (needs /root/reference); the result is committed.

The per-game lists restate the literal asset paths in the reference sources:
  coinrun  games/coinrun/coinrun.cpp:60-110, tilemap.cpp:3-38, common_systems.cpp:107-119,280-282
  maze     games/maze/maze.cpp:62-72, tilemap.cpp:20-29, common_systems.cpp:65-67
"""
import argparse
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, "..", "procgen2_amd", "assets")

GROUND = ["Dirt", "Grass", "Planet", "Sand", "Snow", "Stone"]
WALKERS = ["slimeBlock", "slimePurple", "slimeBlue", "slimeGreen", "mouse", "snail", "ladybug", "wormGreen",
           "wormPink"]
CRATES = ["boxCrate", "boxCrate_double", "boxCrate_single", "boxCrate_warning"]
ALIENS = ["Beige", "Blue", "Green", "Pink", "Yellow"]

PLATFORM_BACKDROPS = [
    "platform_backgrounds/alien_bg.png", "platform_backgrounds/another_world_bg.png",
    "platform_backgrounds/back_cave.png", "platform_backgrounds/caverns.png",
    "platform_backgrounds/cyberpunk_bg.png", "platform_backgrounds/parallax_forest.png",
    "platform_backgrounds/scifi_bg.png", "platform_backgrounds/scifi2_bg.png",
    "platform_backgrounds/living_tissue_bg.png", "platform_backgrounds/airadventurelevel1.png",
    "platform_backgrounds/airadventurelevel2.png", "platform_backgrounds/airadventurelevel3.png",
    "platform_backgrounds/airadventurelevel4.png", "platform_backgrounds/cave_background.png",
    "platform_backgrounds/blue_desert.png", "platform_backgrounds/blue_grass.png",
    "platform_backgrounds/blue_land.png", "platform_backgrounds/blue_shroom.png",
    "platform_backgrounds/colored_desert.png", "platform_backgrounds/colored_grass.png",
    "platform_backgrounds/colored_land.png", "platform_backgrounds/colored_shroom.png",
    "platform_backgrounds/landscape1.png", "platform_backgrounds/landscape2.png",
    "platform_backgrounds/landscape3.png", "platform_backgrounds/landscape4.png",
] + ["platform_backgrounds/battleback%d.png" % i for i in range(1, 11)] + [
    "platform_backgrounds/sunrise.png",
] + ["platform_backgrounds_2/%s%d.png" % (k, i) for k in ("beach", "fantasy", "candy") for i in range(1, 5)]

TOPDOWN_BACKDROPS = ["topdown_backgrounds/floortiles.png"] + [
    "topdown_backgrounds/backgrounddetailed%d.png" % i for i in range(1, 9)]


def coinrun():
    out = []
    for t in GROUND:
        out += ["kenney/Ground/%s/%sMid.png" % (t, t.lower()), "kenney/Ground/%s/%sCenter.png" % (t, t.lower())]
    out += ["kenney/Tiles/lavaTop_low.png", "kenney/Tiles/lava.png"]
    out += ["kenney/Tiles/%s.png" % c for c in CRATES]
    for w in WALKERS:
        out += ["kenney/Enemies/%s.png" % w, "kenney/Enemies/%s_move.png" % w]
    out += ["kenney/Enemies/sawHalf.png", "kenney/Enemies/sawHalf_move.png", "kenney/Items/coinGold.png",
            "misc_assets/iconCircle_white.png"]
    for a in ALIENS:
        out += ["kenney/Players/128x256/%s/alien%s_%s.png" % (a, a, s) for s in ("stand", "jump", "walk1", "walk2")]
    return out + PLATFORM_BACKDROPS


def maze():
    return ["kenney/Ground/Sand/sandCenter.png", "misc_assets/cheese.png",
            "kenney/Enemies/mouse_move.png"] + TOPDOWN_BACKDROPS


def bossfight():
    # games/bossfight/bossfight.cpp:54-79, common_systems.cpp:52-68,474-488
    space = ["deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01", "eye_nebula_01",
             "deep_sky_01", "space_nebula_01", "Background-1", "Background-2", "Background-3", "Background-4",
             "parallax-space-backgound"]
    misc = ["spaceMeteors_001", "spaceMeteors_002", "spaceMeteors_003", "spaceMeteors_004", "meteorGrey_big1",
            "meteorGrey_big2", "meteorGrey_big3", "meteorGrey_big4", "enemyShipBlack1", "enemyShipBlue2",
            "enemyShipGreen3", "enemyShipRed4", "playerShip1_blue", "playerShip1_green", "playerShip2_orange",
            "playerShip3_red", "laserGreen14", "laserRed11", "laserBlue09", "shield2"] + \
           ["explosion%d" % i for i in range(1, 6)]
    return ["space_backgrounds/%s.png" % n for n in space] + ["misc_assets/%s.png" % n for n in misc]


def climber():
    # games/climber/climber.cpp:60-71, tilemap.cpp:3-27, common_systems.cpp:170-182
    tiles = ["tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06", "tileBlue_08", "tileGreen_08",
             "tileYellow_09", "tileBrown_09", "enemySwimming_1", "enemySwimming_2"]
    players = ["player%s_%s" % (c, p) for c in ("Blue", "Green", "Grey", "Red")
               for p in ("stand", "walk4", "walk1", "walk2")]
    backs = ["platform_backgrounds/alien_bg.png", "platform_backgrounds/another_world_bg.png"] + \
            ["platform_backgrounds_2/%s%d.png" % (k, i) for k in ("fantasy", "candy") for i in range(1, 5)]
    return ["platformer/%s.png" % n for n in tiles + players] + ["misc_assets/yellowCrystal.png"] + backs


def caveflyer():
    # games/caveflyer/caveflyer.cpp:58-72, tilemap.cpp:6-20, common_systems.cpp:77-88, :329-331
    misc = ["groundA", "meteorBrown_big1", "ufoRed2", "enemyShipBlue4", "ufoGreen2", "playerShip1_red", "laserBlue02",
            "towerDefense_tile295"] + ["explosion%d" % i for i in range(1, 6)]
    space = ["deep_space_01", "spacegen_01", "milky_way_01", "ez_space_lite_01", "meyespace_v1_01", "eye_nebula_01",
             "deep_sky_01", "space_nebula_01", "Background-1", "Background-2", "Background-3", "Background-4",
             "parallax-space-backgound"]
    return ["misc_assets/%s.png" % n for n in misc] + ["space_backgrounds/%s.png" % n for n in space]


def chaser():
    # games/chaser/chaser.cpp:57-67, tilemap.cpp:6-16, common_systems.cpp:108-115, :297-299
    misc = ["tileStone_slope", "yellowCrystal", "enemySpikey_1b", "enemyFlying_1", "enemyFlying_2", "enemyFlying_3",
            "enemyWalking_1b", "enemyFloating_1b"]
    floors = ["floortiles"] + ["backgrounddetailed%d" % i for i in range(1, 9)]
    return ["misc_assets/%s.png" % n for n in misc] + ["custom/chaser_point.png"] + \
           ["topdown_backgrounds/%s.png" % n for n in floors]


def jumper():
    # games/jumper/jumper.cpp:59-109 (= coinrun's backdrops), :297-299, tilemap.cpp:13-25, common_systems.cpp:50-55, :250
    tiles = ["tileBlue_05", "tileGreen_05", "tileYellow_06", "tileBrown_06", "tileBlue_08", "tileGreen_08",
             "tileYellow_09", "tileBrown_09"]
    misc = ["spikeMan_stand", "carrot", "bunny2_ready", "bunny2_jump", "bunny2_walk1", "bunny2_walk2", "iconCircle_white"]
    custom = ["jumper_compass_circle", "jumper_compass_needle", "jumper_compass_bar"]
    backs = [p for p in coinrun() if p.startswith("platform_backgrounds")]
    return ["platformer/%s.png" % n for n in tiles] + ["misc_assets/%s.png" % n for n in misc] + \
           ["custom/%s.png" % n for n in custom] + backs


GAMES = {"coinrun": coinrun, "maze": maze, "bossfight": bossfight, "climber": climber, "caveflyer": caveflyer,
         "chaser": chaser, "jumper": jumper}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("games", nargs="*", default=sorted(GAMES))
    args = ap.parse_args()
    n = 0
    for g in args.games:
        for rel in GAMES[g]():
            src = os.path.join(args.reference, "assets", rel)
            dst = os.path.join(DEST, rel)
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            if not os.path.exists(dst):
                shutil.copyfile(src, dst)
                n += 1
    print("copied %d files into %s" % (n, os.path.normpath(DEST)))


if __name__ == "__main__":
    main()
