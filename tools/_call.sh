bash tools/ab_trees.sh 3
