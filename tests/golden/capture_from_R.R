# capture_from_R.R -- run on ANY box with R (> 4.3), fields and terra installed:
#
#     Rscript tests/golden/capture_from_R.R          (from the repository root)
#
# It writes tests/golden/r_capture/*.csv; tests/test_r_capture.py picks them up (and is skipped while they are absent).
# This is the one route by which this repository's parity claim can be pinned against the reference's own arithmetic
# (SURVEY.md section 8c, last row): the build container and the GPU box have no R, so the files cannot be made there.
#
# What is captured, for the three station sets of tests/golden/r_inputs/ (the same data as tests/golden/tps_*.npz):
#   * fields::Tps(x, Y) with GCV lambda (the call of V73:722 / V73:751) and with the fixture's fixed lambda:
#     $c, $d, $lambda, $eff.df, $transform$x.center / $x.scale, the GCV grid end points gcv.Krig used;
#   * predict() of both fits on the fixture's 48 x 64 grid of cell centres (what terra::interpolate evaluates,
#     V73:726 / V73:753);
#   * terra::crop windows of the Step-3 tile boxes (V73:656-681, 695-699, 728) for the grids of SURVEY.md G6, as
#     1-based (row0, row1, col0, col1) of the cropped raster inside the parent -- the integer bookkeeping the library
#     reproduces bit for bit.
#   * (where kernlab / nnet are installed) kernlab::ksvm and nnet::nnet fits of tests/golden/r_inputs/learn_fit.csv with
#     the RNG-dependent inputs fixed (sigma; initial weights), and their predict() values.
#   * (where gbm / randomForest / kernlab are installed) gbm, randomForest and ksvm fits on 300 stations with their structures in
#     the loaders' flat layout and their predict() over a 48 x 64 crop of the bundled TWI / slope overviews (round 4).
# sessionInfo() is stored next to the numbers.
suppressPackageStartupMessages({ library(fields); library(terra) })
inp <- file.path("tests", "golden", "r_inputs")
out <- file.path("tests", "golden", "r_capture")
dir.create(out, showWarnings = FALSE, recursive = TRUE)
w <- function(x, name) write.table(format(x, digits = 17), file.path(out, name), sep = ",", quote = FALSE,
                                   row.names = FALSE, col.names = FALSE)

for (name in c("tps_synth12", "tps_synth200", "tps_sampling813")) {
  hdr <- readLines(file.path(inp, paste0(name, ".csv")), n = 1)
  kv <- regmatches(hdr, gregexpr("[a-z]+=[-0-9.e+]+", hdr))[[1]]
  par <- setNames(as.numeric(sub(".*=", "", kv)), sub("=.*", "", kv))
  tab <- read.csv(file.path(inp, paste0(name, ".csv")), comment.char = "#")
  x <- as.matrix(tab[, c("x", "y")]); Y <- tab$resid
  xs <- par["xmin"] + (seq_len(par["ncol"]) - 0.5) * par["res"]        # xFromCol
  ys <- par["ymax"] - (seq_len(par["nrow"]) - 0.5) * par["res"]        # yFromRow
  grid <- as.matrix(expand.grid(x = xs, y = ys))                          # terra cell order: row-major from the NW cell
  for (mode in c("gcv", "fixed")) {
    fit <- if (mode == "gcv") fields::Tps(x, Y) else fields::Tps(x, Y, lambda = par["lam"])
    tag <- paste0(name, "_", mode)
    w(fit$c, paste0(tag, "_c.csv")); w(fit$d, paste0(tag, "_d.csv"))
    w(c(lambda = fit$lambda, eff.df = fit$eff.df, n = nrow(fit$knots), N = length(Y)), paste0(tag, "_scalars.csv"))
    w(rbind(fit$transform$x.center, fit$transform$x.scale), paste0(tag, "_transform.csv"))
    if (!is.null(fit$gcv.grid)) w(as.matrix(fit$gcv.grid[, c("lambda", "trA", "GCV")]), paste0(tag, "_gcvgrid.csv"))
    if (!is.null(fit$lambda.est)) w(as.matrix(fit$lambda.est), paste0(tag, "_lambdaest.csv"))
    w(matrix(predict(fit, grid), nrow = par["nrow"], ncol = par["ncol"], byrow = TRUE), paste0(tag, "_surface.csv"))
  }
}

# ---- Step-3 tile windows (V73:649-681 tile boxes; V73:699 fit crop, V73:728 keep crop) -------------------------------
tile_windows <- function(nrow, ncol, xmin = -78, ymax = -5, res = 1 / 1200) {
  r <- terra::rast(nrows = nrow, ncols = ncol, xmin = xmin, xmax = xmin + ncol * res, ymin = ymax - nrow * res, ymax = ymax)
  nRx <- ceiling(nrow / 1500); nCx <- ceiling(ncol / 1500)
  e <- terra::ext(r)
  longDist <- (e[2] - e[1]) / nCx; latDist <- (e[4] - e[3]) / nRx
  rows <- NULL
  for (j in seq_len(nRx)) for (h in seq_len(nCx)) {                       # row-major from the south-west, V73:664-681
    cell <- c(e[1] + longDist * (h - 1), e[1] + longDist * h, e[3] + latDist * (j - 1), e[3] + latDist * j)
    for (ov in c(0.2, 0.025)) {                                           # fit box, keep box
      b <- terra::ext(cell[1] - ov * longDist, cell[2] + ov * longDist, cell[3] - ov * latDist, cell[4] + ov * latDist)
      cr <- terra::crop(r, b)                                             # for the keep box the reference crops the FIT raster:
      ce <- terra::ext(cr)                                                # the library intersects the two windows
      rows <- rbind(rows, c(nrow, ncol, j, h, ov,
                            terra::rowFromY(r, ce[4] - 0.5 * res), terra::rowFromY(r, ce[3] + 0.5 * res),
                            terra::colFromX(r, ce[1] + 0.5 * res), terra::colFromX(r, ce[2] - 0.5 * res)))
    }
  }
  rows
}
g6 <- rbind(tile_windows(1500, 1500), tile_windows(1501, 1501), tile_windows(2476, 3264), tile_windows(2000, 2000),
            tile_windows(10000, 10000))
colnames(g6) <- c("nrow", "ncol", "tile_row", "tile_col", "overlap", "row0", "row1", "col0", "col1")
write.csv(g6, file.path(out, "step3_tile_windows.csv"), row.names = FALSE, quote = FALSE)
# ---- learner fits (SURVEY.md 8f rank 4), only where kernlab / nnet are installed ---------------------------------------
# tests/golden/r_inputs/learn_fit.csv: resp + 4 predictors; sigma is fixed (ksvm's automatic value is drawn by sigest()),
# the initial nnet weights are given (nnet would draw runif(-0.7, 0.7)).  Captured: what mhs_svr_fit / mhs_nnet_fit and
# the two predict() evaluators must reproduce.
lf <- read.csv(file.path(inp, "learn_fit.csv"), comment.char = "#", header = FALSE, skip = 2,
               col.names = c("resp", "a", "b", "LONG", "LAT"))
sigma <- 0.25
if (requireNamespace("kernlab", quietly = TRUE)) {
  sv <- kernlab::ksvm(resp ~ a + b + LONG + LAT, data = lf, kpar = list(sigma = sigma))       # V73:560 + fixed sigma
  beta <- numeric(nrow(lf)); beta[kernlab::alphaindex(sv)] <- unlist(kernlab::coef(sv))
  w(beta, "learn_ksvm_beta.csv")
  w(c(b = kernlab::b(sv), sigma = sigma, nSV = kernlab::nSV(sv)), "learn_ksvm_scalars.csv")
  sc <- kernlab::scaling(sv)
  w(rbind(sc$x.scale$`scaled:center`, sc$x.scale$`scaled:scale`), "learn_ksvm_xscale.csv")
  w(c(sc$y.scale$`scaled:center`, sc$y.scale$`scaled:scale`), "learn_ksvm_yscale.csv")
  w(as.numeric(kernlab::predict(sv, lf)), "learn_ksvm_predict.csv")
}
if (requireNamespace("nnet", quietly = TRUE)) {
  nn.in <- lf
  mn <- min(nn.in$resp); nn.in$resp <- nn.in$resp - mn; mx <- max(nn.in$resp); nn.in$resp <- nn.in$resp / mx     # V73:455-459
  w0 <- scan(file.path(inp, "learn_fit_wts0.csv"), quiet = TRUE)
  nn <- nnet::nnet(resp ~ a + b + LONG + LAT, data = nn.in, size = 10, linout = TRUE, maxit = 10000, Wts = w0, trace = FALSE)
  w(nn$wts, "learn_nnet_wts.csv")
  w(c(value = nn$value, convergence = nn$convergence, min.resp = mn, max2.resp = mx), "learn_nnet_scalars.csv")
  w(as.numeric(predict(nn, nn.in)) * mx + mn, "learn_nnet_predict.csv")                       # V73:468-470
  nn25 <- nnet::nnet(resp ~ a + b + LONG + LAT, data = nn.in, size = 10, linout = TRUE, maxit = 25, Wts = w0, trace = FALSE)
  w(nn25$wts, "learn_nnet_wts_maxit25.csv")
}
# ---- tree / kernel members predicted over a raster crop (SURVEY.md 8a-3, 8a-4, 8a-8), where gbm / randomForest / kernlab are installed --
# tests/golden/r_inputs/members_stations.csv: resp + alt, slope, TWI, LONG, LAT at 300 stations; members_crop.csv: the same five
# predictors at the 48 x 64 cells of a crop of the bundled TWI / slope overviews (terra cell order).  For each package the
# FITTED STRUCTURE is written in the flat layout the library's loaders take (include/machisplin_hip.h: mhs_gbm_load,
# mhs_rf_load, mhs_svr_load) together with the package's own predict() over the crop: the test loads the structure and must
# reproduce the prediction, which pins the evaluators (not the trainers) against the packages the reference calls at
# V73:497 (gbm), V73:521-523 (randomForest) and V73:582-584 (ksvm).
st <- read.csv(file.path(inp, "members_stations.csv"), comment.char = "#", header = FALSE, skip = 2,
               col.names = c("resp", "alt", "slope", "TWI", "LONG", "LAT"))
crop <- read.csv(file.path(inp, "members_crop.csv"), comment.char = "#", header = FALSE, skip = 2,
                 col.names = c("alt", "slope", "TWI", "LONG", "LAT"))
if (requireNamespace("gbm", quietly = TRUE)) {
  set.seed(1)
  gb <- gbm::gbm(resp ~ alt + slope + TWI + LONG + LAT, data = st, distribution = "gaussian", n.trees = 300,
                 interaction.depth = 5, shrinkage = 0.01, bag.fraction = 0.75, n.minobsinnode = 5, verbose = FALSE)
  nt <- 250                                                                  # "best.trees" < n.trees, as V73:497 cuts the object
  rows <- NULL
  for (i in seq_len(nt)) {
    tr <- gb$trees[[i]]                                                      # SplitVar, SplitCodePred, LeftNode, RightNode, MissingNode, ...
    rows <- rbind(rows, cbind(i - 1, tr[[1]], tr[[2]], tr[[3]], tr[[4]], tr[[5]]))
  }
  w(rows, "members_gbm_nodes.csv")                                           # tree, SplitVar (0-based, -1 terminal), SplitCodePred, Left, Right, Missing
  w(c(initF = gb$initF, n.trees = nt, p = 5), "members_gbm_scalars.csv")
  w(as.numeric(gbm::predict.gbm(gb, crop, n.trees = nt, type = "response")), "members_gbm_predict.csv")
}
if (requireNamespace("randomForest", quietly = TRUE)) {
  set.seed(2)
  rf <- randomForest::randomForest(resp ~ alt + slope + TWI + LONG + LAT, data = st, ntree = 60)   # V73:517 passes no ntree; 60 keeps the file small
  f <- rf$forest
  rows <- NULL
  for (t in seq_len(f$ntree)) {
    k <- seq_len(f$ndbigtree[t])
    rows <- rbind(rows, cbind(t - 1, f$leftDaughter[k, t], f$rightDaughter[k, t], f$nodestatus[k, t], f$bestvar[k, t],
                              f$xbestsplit[k, t], f$nodepred[k, t]))
  }
  w(rows, "members_rf_nodes.csv")                                            # tree, leftDaughter, rightDaughter (1-based), nodestatus, bestvar (1-based), xbestsplit, nodepred
  w(as.numeric(predict(rf, crop, type = "response")), "members_rf_predict.csv")
}
if (requireNamespace("kernlab", quietly = TRUE)) {
  sigma <- 0.2
  sv <- kernlab::ksvm(resp ~ alt + slope + TWI + LONG + LAT, data = st, kpar = list(sigma = sigma))   # eps-svr, rbfdot, scaled = TRUE
  sc <- kernlab::scaling(sv)
  w(cbind(unlist(kernlab::coef(sv)), kernlab::xmatrix(sv)), "members_ksvm_sv.csv")      # alpha, then the scaled support vector (5 columns)
  w(c(b = kernlab::b(sv), sigma = sigma, nSV = kernlab::nSV(sv)), "members_ksvm_scalars.csv")
  w(rbind(sc$x.scale$`scaled:center`, sc$x.scale$`scaled:scale`), "members_ksvm_xscale.csv")
  w(c(sc$y.scale$`scaled:center`, sc$y.scale$`scaled:scale`), "members_ksvm_yscale.csv")
  w(as.numeric(kernlab::predict(sv, crop)), "members_ksvm_predict.csv")
}
writeLines(capture.output(sessionInfo()), file.path(out, "sessionInfo.txt"))
cat("wrote", length(list.files(out)), "files to", out, "\n")
