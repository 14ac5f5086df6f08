graph [
  directed 1
  node [
    id 0
    label "0"
  ]
  node [
    id 1
    label "1"
  ]
  node [
    id 2
    label "2"
  ]
  node [
    id 3
    label "3"
  ]
  node [
    id 4
    label "4"
  ]
  node [
    id 5
    label "5"
  ]
  node [
    id 6
    label "6"
  ]
  node [
    id 7
    label "7"
  ]
  node [
    id 8
    label "8"
  ]
  node [
    id 9
    label "9"
  ]
  node [
    id 10
    label "10"
  ]
  node [
    id 11
    label "11"
  ]
  node [
    id 12
    label "12"
  ]
  node [
    id 13
    label "13"
  ]
  edge [
    source 0
    target 1
    capacity 2879
  ]
  edge [
    source 0
    target 2
    capacity 1307
  ]
  edge [
    source 0
    target 7
    capacity 3100
  ]
  edge [
    source 1
    target 2
    capacity 3257
  ]
  edge [
    source 1
    target 3
    capacity 2210
  ]
  edge [
    source 1
    target 0
    capacity 2879
  ]
  edge [
    source 2
    target 5
    capacity 1596
  ]
  edge [
    source 2
    target 0
    capacity 1307
  ]
  edge [
    source 2
    target 1
    capacity 3257
  ]
  edge [
    source 3
    target 4
    capacity 3315
  ]
  edge [
    source 3
    target 10
    capacity 4438
  ]
  edge [
    source 3
    target 1
    capacity 2210
  ]
  edge [
    source 4
    target 5
    capacity 3687
  ]
  edge [
    source 4
    target 6
    capacity 3472
  ]
  edge [
    source 4
    target 3
    capacity 3315
  ]
  edge [
    source 5
    target 9
    capacity 1919
  ]
  edge [
    source 5
    target 12
    capacity 1274
  ]
  edge [
    source 5
    target 2
    capacity 1596
  ]
  edge [
    source 5
    target 4
    capacity 3687
  ]
  edge [
    source 6
    target 7
    capacity 1819
  ]
  edge [
    source 6
    target 4
    capacity 3472
  ]
  edge [
    source 7
    target 8
    capacity 2459
  ]
  edge [
    source 7
    target 0
    capacity 3100
  ]
  edge [
    source 7
    target 6
    capacity 1819
  ]
  edge [
    source 8
    target 9
    capacity 4255
  ]
  edge [
    source 8
    target 11
    capacity 1944
  ]
  edge [
    source 8
    target 13
    capacity 4962
  ]
  edge [
    source 8
    target 7
    capacity 2459
  ]
  edge [
    source 9
    target 5
    capacity 1919
  ]
  edge [
    source 9
    target 8
    capacity 4255
  ]
  edge [
    source 10
    target 11
    capacity 1347
  ]
  edge [
    source 10
    target 13
    capacity 3230
  ]
  edge [
    source 10
    target 3
    capacity 4438
  ]
  edge [
    source 11
    target 12
    capacity 2973
  ]
  edge [
    source 11
    target 8
    capacity 1944
  ]
  edge [
    source 11
    target 10
    capacity 1347
  ]
  edge [
    source 12
    target 13
    capacity 1851
  ]
  edge [
    source 12
    target 5
    capacity 1274
  ]
  edge [
    source 12
    target 11
    capacity 2973
  ]
  edge [
    source 13
    target 8
    capacity 4962
  ]
  edge [
    source 13
    target 10
    capacity 3230
  ]
  edge [
    source 13
    target 12
    capacity 1851
  ]
]
