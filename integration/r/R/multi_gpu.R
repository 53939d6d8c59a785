# R-side multi-GPU driver for machisplin.tiles.* runs (UNTESTED HERE: no R in the build image; the same scheme runs in
# Python as machisplin_amd/sharded.py::TileShardedMltps and is covered there).  The reference's only parallelism ever was
# snowfall's sfLapply over response layers (old/...V69.R:964); its current README runs machisplin.mltps once per tile of
# machisplin.tiles.create and merges with machisplin.tiles.merge (README.md:157-215).  Here the (tile, layer) units of such a
# run are dealt over the node's GPUs: one R worker process per GPU (parallel::makeCluster, type "PSOCK" -- a HIP context must
# not be forked), each worker binds its device with options(machisplin.device = rank) before the backend initialises, runs
# machisplin.mltps for its units with options(machisplin.backend = "hip") and returns the final rasters' values; the master
# merges every layer with machisplin.tiles.merge.  The library has no collective inside: planes travel through R's own
# serialisation, 8 bytes per cell and unit (cfg4: 48 units x 25 M cells = 9.7 GB over the workers' sockets -- minutes of R
# time against the step's second on the GPUs: write the units to GeoTIFF on the workers instead when that matters).
#
#   tiles  <- machisplin.tiles.create(int.values, covar.ras, out.ncol = 2, out.nrow = 2, feather.d = 50)
#   finals <- mhs_tiles_mltps(tiles, n.gpus = 4)            # list over response layers of lists over tiles
#   bio1   <- machisplin.tiles.merge(finals[[1]], in.ncol = 2, in.nrow = 3)

mhs_unit_owner <- function(tile, layer, n.tiles, n.gpus) {
  u <- (layer - 1L) * n.tiles + (tile - 1L)                  # layer-major numbering, as sharded.unit_owner
  c(rank = u %% n.gpus, slot = u %/% n.gpus)
}

mhs_tiles_mltps <- function(tiles, n.gpus = 1L, ...) {
  n.tiles <- length(tiles$dat)
  n.layers <- ncol(tiles$dat[[1]]) - 2L                      # long, lat, then the response layers
  units <- expand.grid(tile = seq_len(n.tiles), layer = seq_len(n.layers))
  units$rank <- mapply(function(t, l) mhs_unit_owner(t, l, n.tiles, n.gpus)[["rank"]], units$tile, units$layer)
  cl <- parallel::makeCluster(n.gpus, type = "PSOCK")
  on.exit(parallel::stopCluster(cl))
  parallel::clusterApply(cl, seq_len(n.gpus) - 1L, function(rank) {
    options(machisplin.backend = "hip", machisplin.device = rank)
    library(MACHISPLIN)
    invisible(.Call("mhsr_init", as.integer(rank)))
  })
  run.rank <- function(mine, tiles, ...) {
    # a worker keeps ONE tile's rasters in memory at a time; its layers share the Step-3 reductions (mhsr_tps_reduction_cache)
    out <- vector("list", nrow(mine))
    for (t in unique(mine$tile)) {
      ras <- terra::rast(tiles$rast[[t]])
      .Call("mhsr_tps_reduction_cache", 1L)
      for (k in which(mine$tile == t)) {
        l <- mine$layer[k]
        dat <- tiles$dat[[t]][, c(1, 2, 2 + l)]
        res <- machisplin.mltps(int.values = dat, covar.ras = ras, n.cores = 1, ...)
        out[[k]] <- list(tile = t, layer = l, values = terra::values(res[[1]]$final), geom = c(nrow(ras), ncol(ras)))
      }
      .Call("mhsr_tps_reduction_cache", 0L)
    }
    out
  }
  parts <- parallel::clusterApply(cl, seq_len(n.gpus) - 1L, function(rank, units, tiles, run.rank, ...)
    run.rank(units[units$rank == rank, , drop = FALSE], tiles, ...), units, tiles, run.rank, ...)
  finals <- lapply(seq_len(n.layers), function(l) vector("list", n.tiles))
  for (p in parts) for (u in p) {
    r <- terra::rast(tiles$rast[[u$tile]][[1]])
    finals[[u$layer]][[u$tile]] <- terra::setValues(r, u$values)
  }
  finals
}
